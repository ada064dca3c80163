"""Experiment: how much of the latency-bound step is recovered when TWO independent half-batch steps run concurrently
(two TrainStep instances of batch B/2 on two streams, separate graphs) instead of one step of batch B?
Usage: python tools/exp_dual_chain.py [B]"""
import sys, time
sys.path.insert(0, ".")
import torch
from camradepth_amd import synth
from camradepth_amd.model import CamRaDepth
from camradepth_amd.trainer import TrainStep

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8


def make(b, seed):
    m = CamRaDepth(input_channels=7, seed=0).cuda().train()
    ts = TrainStep(m, b, 256, 416, use_graph=True)
    ts.set_batch({k: v.cuda() for k, v in synth.make_batch(b, 256, 416, seed=seed).items()})
    for _ in range(3):
        ts.step()
    torch.cuda.synchronize()
    return ts


def timeit(fn, n=20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


full = make(B, 1)
print(f"one step of batch {B}: {timeit(full.step):.2f} ms")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
with torch.cuda.stream(s1):
    a = make(B // 2, 2)
with torch.cuda.stream(s2):
    b = make(B // 2, 3)
print(f"one step of batch {B // 2} alone: {timeit(lambda: (torch.cuda.current_stream().wait_stream(s1), a.step())):.2f} ms")


def both():
    with torch.cuda.stream(s1):
        a.step()
    with torch.cuda.stream(s2):
        b.step()


print(f"two concurrent steps of batch {B // 2}: {timeit(both):.2f} ms")
