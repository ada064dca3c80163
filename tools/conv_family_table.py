#!/usr/bin/env python3
"""Per-launch table of the 3x3 convolution family of one C2 training step (forward and data-gradient launches in plan order):
kernel label, shape, algorithmic GFLOP, time of 20 back-to-back graph replays of that one launch, TFLOP/s.
    python tools/conv_family_table.py [--batch 8]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import camradepth_amd.lib as L
from camradepth_amd import synth
from camradepth_amd.model import CamRaDepth
from camradepth_amd.trainer import TrainStep

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--family", default="k_conv3x3")
a = ap.parse_args()
m = CamRaDepth(input_channels=7).cuda().train()
ts = TrainStep(m, a.batch, 256, 416, lr=6e-5)
ts.start_epoch()
ts.set_batch({k: v.cuda() for k, v in synth.make_batch(a.batch, 256, 416, seed=1).items()})
for _ in range(2):
    ts.step()
torch.cuda.synchronize()
plan = ts.plan
rows = []
s = torch.cuda.Stream()
for where, ops in (("fwd", plan.fwd), ("bwd", plan.bwd)):
    for op in ops:
        if op.fn is None or op.meta is None or not op.meta["kernel"].startswith(a.family):
            continue
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            op.fn(*op.args, L.stream())
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                for _ in range(20):
                    op.fn(*op.args, L.stream())
            g.replay()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            g.replay()
            e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        rows.append((where, op.meta["kernel"], op.meta.get("shape", ""), op.meta["flops"] / 1e9, us))
tot_f = sum(r[3] for r in rows); tot_t = sum(r[4] for r in rows)
for r in rows:
    print(f"{r[0]}  {r[1]:<36s} {str(r[2]):<44s} {r[3]:8.1f} GFLOP {r[4]:8.1f} us {r[3] / r[4] * 1e3:7.0f} TFLOP/s")
print(f"total {tot_f:.0f} GFLOP, {tot_t / 1e3:.3f} ms, {tot_f / tot_t * 1e3:.0f} TFLOP/s = {tot_f / tot_t * 1e3 / 2500:.3f} of the dense bf16 peak")
