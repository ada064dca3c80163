import sys, collections
sys.path.insert(0, ".")
import torch
from camradepth_amd.model import CamRaDepth
from camradepth_amd.inference import InferenceGraph
from camradepth_amd import lib as L
m = CamRaDepth(input_channels=7, seed=0).cuda()
ig = InferenceGraph(m, 8, 256, 416)
plan = ig.plan
st = L.stream()
hist = collections.Counter(op.name for op in plan.fwd if op.fn is not None)
print(sum(hist.values()), "launch records:", dict(hist))
# eager per-op timing
tot = collections.defaultdict(float)
plan.forward(); torch.cuda.synchronize()
for rep in range(3):
    evs = []
    plan.zf_arena.zero_()
    for op in plan.fwd:
        if op.fn is None: continue
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); op.fn(*op.args, st); e1.record()
        evs.append((op.name, e0, e1))
    torch.cuda.synchronize()
    if rep == 2:
        for n, e0, e1 in evs: tot[n] += e0.elapsed_time(e1)
for k, v in sorted(tot.items(), key=lambda kv: -kv[1]): print(f"{k:32s} {v:7.3f} ms  x{hist[k]}")
print("sum", sum(tot.values()))
# by stage marks
marks = plan.fwd_marks + [("end", len(plan.fwd))]
