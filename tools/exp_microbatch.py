"""Experiment: does running two half-batch plans concurrently on two streams beat one full-batch plan? (timing only)"""
import sys, torch
sys.path.insert(0, ".")
from camradepth_amd.config import ModelConfig
from camradepth_amd.engine import Plan
from camradepth_amd.model import CamRaDepth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
nm = int(sys.argv[2]) if len(sys.argv) > 2 else 2
m = CamRaDepth(input_channels=7).cuda().train()
m._ensure_grad_views()


def fb(p):
    p.forward(); p.backward()


def timed(fn, n=10):
    g = torch.cuda.CUDAGraph()
    fn(); torch.cuda.synchronize()
    with torch.cuda.graph(g):
        fn()
    for _ in range(3): g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


full = Plan(m, B, 256, 416, True)
print(f"one plan B={B}: {timed(lambda: fb(full)):.2f} ms fwd+bwd")
halves = [Plan(m, B // nm, 256, 416, True) for _ in range(nm)]
side = [torch.cuda.Stream() for _ in range(nm - 1)]


def both():
    cur = torch.cuda.current_stream()
    for s in side: s.wait_stream(cur)
    for p, s in zip(halves[1:], side):
        with torch.cuda.stream(s):
            fb(p)
    fb(halves[0])
    for s in side: cur.wait_stream(s)


def serial():
    for p in halves: fb(p)


print(f"{nm} plans B={B // nm} serial: {timed(serial):.2f} ms")
print(f"{nm} plans B={B // nm} concurrent: {timed(both):.2f} ms")

# variant: one graph per half, replayed on two different streams
gs = []
for p in halves:
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fb(p)
    gs.append(g)
streams = [torch.cuda.Stream() for _ in halves]
torch.cuda.synchronize()
def launch_all():
    for g, s in zip(gs, streams):
        with torch.cuda.stream(s):
            g.replay()
for _ in range(3): launch_all()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(10): launch_all()
torch.cuda.synchronize()
print(f"{nm} graphs on {nm} streams: {(time.perf_counter() - t0) * 100:.2f} ms")
