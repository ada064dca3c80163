#!/usr/bin/env python3
"""Upper bound of micro-batch pipelining: one B=8 training step against two independent B=4 steps (two models, two
streams) replayed side by side.  GroupNorm is per sample and every cross-sample sum is an order-independent crd_sum_t,
so splitting the batch would not change a bit of the result; this measures what the overlap could buy.
    python tools/exp_split_batch.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from camradepth_amd import synth
from camradepth_amd.model import CamRaDepth
from camradepth_amd.trainer import TrainStep


def make(B):
    torch.manual_seed(0)
    m = CamRaDepth(input_channels=7).cuda().train()
    ts = TrainStep(m, B, 256, 416, lr=6e-5)
    ts.start_epoch()
    ts.set_batch({k: (v.cuda() if torch.is_tensor(v) else v) for k, v in synth.make_batch(B, 256, 416, seed=1).items()})
    return ts


def timed(fn, n=30, w=5):
    for _ in range(w):
        fn()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.time() - t0) / n * 1e3


def main():
    t8 = make(8)
    print("B=8, one step: %.3f ms" % timed(t8.step))
    a, b = make(4), make(4)
    print("B=4, one step: %.3f ms" % timed(a.step))
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()

    def both():
        cur = torch.cuda.current_stream()
        sa.wait_stream(cur); sb.wait_stream(cur)
        with torch.cuda.stream(sa):
            a.step()
        with torch.cuda.stream(sb):
            b.step()
        cur.wait_stream(sa); cur.wait_stream(sb)
    print("2 x B=4 on two streams: %.3f ms per pair" % timed(both))


if __name__ == "__main__":
    main()
