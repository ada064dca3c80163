"""Developer tool: the backward (or forward) launch sequence of ONE encoder block, launch by launch: graph-replayed time of each launch
alone (warm caches: the launch's floor), its kernel, shape and epilogue flags.  Usage: python tools/chain_table.py fwd|bwd first last   (indices into the plan's op list)"""
import sys

import torch

from camradepth_amd import synth, lib as L
from camradepth_amd.model import CamRaDepth
from camradepth_amd.trainer import TrainStep

which = sys.argv[1] if len(sys.argv) > 1 else "bwd"
B = 8
import os
B = int(os.environ.get("CRD_CHAIN_BATCH", B))
model = CamRaDepth(input_channels=7).cuda().train()
batch = synth.make_batch(B, 256, 416, seed=1234)
if os.environ.get("CRD_CHAIN_FP8"):          # config 5: e4m3 forward (+ data gradients with =grad) in decoder stages 3-4
    model.calibrate_fp8(batch["image"].cuda(), train=True, grads=os.environ["CRD_CHAIN_FP8"] == "grad")
ts = TrainStep(model, B, 256, 416, use_graph=False)
ts.set_batch({k: v.cuda() for k, v in batch.items()})
ts.step()
torch.cuda.synchronize()
plan = ts.plan
if plan.fp8_grad_layers:
    plan.fp8_jit = False                      # time the delayed-scaling variant (what TrainStep's graphs replay)
ops = plan.bwd if which == "bwd" else plan.fwd


def describe(op):
    out = []
    for a in op.args:
        if isinstance(a, dict):
            for k in ("red", "accumulate", "acc", "stats", "chan", "res", "bias", "act", "gn_in", "dbias"):
                if a.get(k) is not None and a.get(k) is not False and a.get(k) != 0:
                    out.append(k)
            y = a.get("y")
            if y is not None and getattr(y, "f32", 0):
                out.append("y_f32")
    return ",".join(out)


def time_op(op, reps=20):
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        plan.run_ops([op])
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for _ in range(reps):
                plan.run_ops([op])
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


lo = int(sys.argv[2]) if len(sys.argv) > 2 else 0
hi = int(sys.argv[3]) if len(sys.argv) > 3 else 60
tot, n = 0.0, 0
for i, op in enumerate(ops):
    if not plan.live(op) or not (lo <= i < hi):
        continue
    us = time_op(op)
    tot += us
    n += 1
    m = op.meta or {}
    print(f"{i:5d} {us:7.1f} us  s{op.stream} {op.name:26s} {str(m.get('kernel', '')):24s} {str(m.get('shape', plan.shapes.get(id(op), ''))):46s} {describe(op)}", flush=True)
print(f"sum {tot:.1f} us over {n} launches")
