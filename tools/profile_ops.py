"""Developer tool: per-launch HIP-event timing of one training step (eager), grouped by C-ABI entry point and by op."""
import sys, collections
sys.path.insert(0, ".")
import torch
from camradepth_amd import synth, lib as L
from camradepth_amd.model import CamRaDepth
from camradepth_amd.trainer import TrainStep
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
variant = sys.argv[2] if len(sys.argv) > 2 else "base"
model = CamRaDepth(input_channels=7, supervised_seg=(variant == "supervised_seg")).cuda().train()
ts = TrainStep(model, B, 256, 416, use_graph=False)
batch = synth.make_batch(B, 256, 416, seed=1234)
ts.set_batch({k: v.cuda() for k, v in batch.items()})
for _ in range(2):
    ts.step()
torch.cuda.synchronize()
plan, st = ts.plan, L.stream()
rec = []
for rep in range(3):
    ts._forward_and_loss_partials(); ts._loss_backward(); plan.zb_arena.zero_(); plan.zf_arena.zero_()
    for phase, ops in (("fwd", plan.fwd), ("bwd", plan.bwd)):
        evs = []
        for i, op in enumerate(ops):
            if op.fn is None:
                continue
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); op.fn(*op.args, st); e1.record()
            evs.append((phase, i, op, e0, e1))
        torch.cuda.synchronize()
        if rep == 2:
            rec += [(ph, i, op, e0.elapsed_time(e1)) for ph, i, op, e0, e1 in evs]
tot = sum(r[3] for r in rec)
print(f"total per-op time {tot:.2f} ms over {len(rec)} launches (fwd {sum(r[3] for r in rec if r[0]=='fwd'):.2f}, bwd {sum(r[3] for r in rec if r[0]=='bwd'):.2f})")
by = collections.defaultdict(lambda: [0, 0.0])
for ph, i, op, ms in rec:
    k = op.name + (" " + op.meta["kernel"] if op.meta else "")
    by[k][0] += 1; by[k][1] += ms
for k, v in sorted(by.items(), key=lambda kv: -kv[1][1]):
    print(f"  {v[1]:8.3f} ms {v[0]:5d}x  {k}")
print("top ops:")
for ph, i, op, ms in sorted(rec, key=lambda r: -r[3])[:45]:
    extra = ""
    if op.meta:
        extra = f"{op.meta['shape']}  {op.meta['flops'] / ms / 1e9:.0f} TF/s"
    print(f"  {ms:7.3f} ms {ph} #{i:4d} {op.name} {extra}")

import json
with open("gpurun_out/ops_all.json", "w") as f:
    json.dump([(ph, i, op.name, (op.meta or {}).get("shape", ""), ms) for ph, i, op, ms in rec], f)

# time per model segment
marks = plan.fwd_marks + [("end", len(plan.fwd))]
shift = 2 if len(plan.fwd) > 2 else 0      # two slice copies are inserted at the front after the marks were taken
print("segment      fwd ms   bwd ms   launches")
for (tag, a), (_, b2) in zip(marks, marks[1:]):
    a2, b3 = (a + shift if a > 0 else a), b2 + shift
    f = sum(ms for ph, i, op, ms in rec if ph == "fwd" and a2 <= i < b3)
    seg = [sg for sg in plan.bwd_segments if sg[0] == tag]
    bw = sum(ms for ph, i, op, ms in rec if ph == "bwd" and seg and seg[0][1] <= i < seg[0][2])
    nl = sum(1 for ph, i, op, ms in rec if (ph == "fwd" and a2 <= i < b3) or (ph == "bwd" and seg and seg[0][1] <= i < seg[0][2]))
    print(f"{tag:10s} {f:8.2f} {bw:8.2f}   {nl}")

# average time of every op type per encoder stage (eager event timing: ~5 us of launch path included)
print("op avg us by stage (fwd / bwd):")
tab = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for (tag, a0), (_, b0) in zip(marks, marks[1:]):
    a2, b3 = (a0 + shift if a0 > 0 else a0), b0 + shift
    for ph, i, op, ms in rec:
        if ph == "fwd" and a2 <= i < b3:
            e = tab[("F " + op.name)][tag]; e[0] += 1; e[1] += ms
for tag, a0, b0 in plan.bwd_segments:
    for ph, i, op, ms in rec:
        if ph == "bwd" and a0 <= i < b0:
            e = tab[("B " + op.name)][tag]; e[0] += 1; e[1] += ms
tags = ["enc0", "enc1", "enc2", "enc3", "dec"]
print(f"{'':34s}" + "".join(f"{t:>16s}" for t in tags))
for k in sorted(tab, key=lambda k: -sum(v[1] for v in tab[k].values())):
    row = "".join((f"{tab[k][t][1] / tab[k][t][0] * 1e3:9.1f} x{tab[k][t][0]:<5d}" if tab[k][t][0] else f"{'':16s}") for t in tags)
    print(f"{k:34s}{row}")
