"""Developer tool: the step's floor budget against measured time, per phase and per kernel.

Every launch of the C2 training plan (base, 8 x 7 x 256 x 416) is replayed ALONE from a HIP graph (warm caches: what the launch costs when
nothing else runs -- tools/chain_table.py's measurement) and put next to its floor = max(algorithmic FLOPs at the sustained MFMA rate,
algorithmic bytes at the sustained HBM rate, one dispatch boundary) -- bench.floor_budget's model.  Output: a per-phase table (launches,
MFMA / byte / dependency floor, phase floor, warm time), and the kernels ranked by the time they spend above their floor per step.

    python tools/floor_table.py [--md profiles/rNN_floor_budget.md] [--step-ms 17.9]
"""
import argparse
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from camradepth_amd import synth  # noqa: E402
from camradepth_amd.engine import LATE  # noqa: E402
from camradepth_amd.model import CamRaDepth  # noqa: E402
from camradepth_amd.trainer import TrainStep  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--md", default=None)
ap.add_argument("--step-ms", type=float, default=None, help="measured ms per replayed step of the same build (bench.py), for the last line")
ap.add_argument("--batch", type=int, default=8)
a = ap.parse_args()

B = a.batch
model = CamRaDepth(input_channels=7).cuda().train()
ts = TrainStep(model, B, 256, 416, use_graph=False)
ts.set_batch({k: v.cuda() for k, v in synth.make_batch(B, 256, 416, seed=1234).items()})
ts.step()
torch.cuda.synchronize()
plan = ts.plan


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
    return e0.elapsed_time(e1) / reps          # ms


def floor_of(op):
    fl = float((op.meta or {}).get("flops", 0.0))
    by = float(plan.op_bytes(op))
    n = 1 + (op.meta or {}).get("kernel", "").count("+")
    tm, tb, td = fl / (bench.MFMA_SUSTAINED_TFLOPS * 1e9), by / (bench.HBM_SUSTAINED_TBS * 1e9), n * bench.DISPATCH_FLOOR_US * 1e-3
    return tm, tb, td, max(tm, tb, td), by


marks = plan.fwd_marks + [("end", len(plan.fwd))]
rows = []          # (phase, kernel label, ms, mfma, byte, dep, floor, bytes)
for (name, lo), (_, hi) in zip(marks[:-1], marks[1:]):
    for i, op in enumerate(plan.fwd[:hi] if name == marks[0][0] else plan.fwd[lo:hi]):
        if plan.live(op):
            rows.append(("fwd:" + name, op, time_op(op)) + floor_of(op))
saved_split = plan.split_late
plan.split_late = False
for tag, lo, hi in plan.bwd_segments:
    for op in plan.bwd[lo:hi]:
        if plan.live(op):
            rows.append((("late:" if op.stream == LATE else "bwd:") + tag, op, time_op(op)) + floor_of(op))
plan.split_late = saved_split

out = []
P = collections.OrderedDict()
for ph, op, ms, tm, tb, td, fl, by in rows:
    p = P.setdefault(ph, [0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0])
    p[0] += 1 + (op.meta or {}).get("kernel", "").count("+"); p[1] += tm; p[2] += tb; p[3] += td; p[4] += fl; p[5] += ms; p[6] += by
out.append(f"# Floor budget, C2 training step (base, {B} x 7 x 256 x 416)\n")
out.append(f"Floor of a launch = max(FLOPs / {bench.MFMA_SUSTAINED_TFLOPS:.0f} TFLOP/s, algorithmic bytes / {bench.HBM_SUSTAINED_TBS} TB/s, "
           f"{bench.DISPATCH_FLOOR_US} us); `warm` = the launch replayed alone from a HIP graph (tools/floor_table.py).\n")
out.append("| phase | launches | MFMA floor ms | byte floor ms | dependency floor ms | phase floor ms | warm ms | warm / floor | algorithmic GB |")
out.append("|---|---|---|---|---|---|---|---|---|")
tot = [0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0]
totl = [0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0]
for ph, p in P.items():
    out.append(f"| {ph} | {p[0]} | {p[1]:.3f} | {p[2]:.3f} | {p[3]:.3f} | {p[4]:.3f} | {p[5]:.3f} | {p[5] / p[4]:.2f} | {p[6] / 1e9:.2f} |")
    t = totl if ph.startswith("late:") else tot
    for i in range(7):
        t[i] += p[i]
out.append(f"| **chain total** | {tot[0]} | {tot[1]:.3f} | {tot[2]:.3f} | {tot[3]:.3f} | **{tot[4]:.3f}** | **{tot[5]:.3f}** | {tot[5] / tot[4]:.2f} | {tot[6] / 1e9:.2f} |")
out.append(f"| late stream total | {totl[0]} | {totl[1]:.3f} | {totl[2]:.3f} | {totl[3]:.3f} | {totl[4]:.3f} | {totl[5]:.3f} | {totl[5] / max(totl[4], 1e-9):.2f} | {totl[6] / 1e9:.2f} |")
if a.step_ms:
    out.append(f"\nMeasured replayed step: {a.step_ms:.2f} ms = {a.step_ms / tot[4]:.2f} x the chain floor, {a.step_ms / tot[5]:.2f} x the warm chain "
               f"(the rest: cold caches between dependent launches, late-stream interference, the tail behind the last segment).\n")
K = collections.OrderedDict()
for ph, op, ms, tm, tb, td, fl, by in rows:
    label = (op.meta or {}).get("kernel") or op.name
    label = ("late " if ph.startswith("late:") else "") + label.split("<")[0] + ("<" + label.split("<")[1] if "<" in label else "")
    k = K.setdefault(label, [0, 0.0, 0.0, 0.0, 0.0, 0.0])
    k[0] += 1; k[1] += ms; k[2] += fl; k[3] += by; k[4] += float((op.meta or {}).get("flops", 0.0)); k[5] += td
out.append("\n## Kernels ranked by warm time above their floor (per step)\n")
out.append("| kernel | launches | warm ms | floor ms | above floor ms | GB/s (algorithmic) | TFLOP/s |")
out.append("|---|---|---|---|---|---|---|")
for label, k in sorted(K.items(), key=lambda kv: -(kv[1][1] - kv[1][2]))[:40]:
    out.append(f"| {label} | {k[0]} | {k[1]:.3f} | {k[2]:.3f} | {k[1] - k[2]:.3f} | {k[3] / k[1] / 1e6:.0f} | {k[4] / k[1] / 1e9:.0f} |")
text = "\n".join(out) + "\n"
print(text)
if a.md:
    open(a.md, "w").write(text)
