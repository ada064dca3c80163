"""profiles/rNN_pmc_traffic.json (HBM bytes per kernel from the FETCH_SIZE / WRITE_SIZE counter passes) joined with
profiles/rNN_one_step_kernels.txt (in-step launches and durations from the rocprofv3 trace of one replayed step) ->
profiles/rNN_hbm_bytes_by_kernel.md: where the step's HBM bytes go and at what rate each kernel moves them.
Usage: python tools/bytes_table.py r04"""
import json
import re
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
tr = json.load(open(f"profiles/{tag}_pmc_traffic.json"))
step = {}
wall = None
for line in open(f"profiles/{tag}_one_step_kernels.txt"):
    m = re.match(r"\s*([\d.]+) ms\s+(\d+)x\s+([\d.]+) us\s+(.*)", line)
    if m:
        step[m.group(4).strip()] = (float(m.group(1)), int(m.group(2)), float(m.group(3)))
    elif line.startswith("step wall"):
        wall = float(line.split()[2])
rows = []
for name, v in tr["kernels"].items():
    st = step.get(name)
    if st is None:
        continue
    ms, n, us = st
    f, w = v["fetch_bytes_per_launch"], v["write_bytes_per_launch"]
    iters = 3                                        # the counter passes run bench.py --steps 2 --warmup 1 (eager): three iterations
    per_step = v["launches"] * (f + w) / iters       # (eager and graph-replayed steps launch the same kernels, except the optimizer: one
    rows.append((per_step, name, n, v["launches"] / iters, f, w, ms))          #  launch per step eagerly, one per gradient bucket in the replayed step)
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
out = [f"# HBM bytes per kernel, training step (B = 8, 256 x 416) -- {tag}", "",
       f"Counters: `profiles/{tag}_pmc_traffic.json` ({tr['source']}); launches and durations: `profiles/{tag}_one_step_kernels.txt` "
       f"(one replayed step under the kernel trace, step wall {wall} ms).  Kernels present in both files: {tot / 1e9:.1f} GB per step.", "",
       "| kernel | launches / step (trace) | launches / step (counter run) | fetched MB / launch | written MB / launch | GB / step | share | in-step ms | TB/s |",
       "|---|---|---|---|---|---|---|---|---|"]
for b, name, n, nl, f, w, ms in rows[:45]:
    out.append(f"| `{name}` | {n} | {nl:.0f} | {f / 1e6:.1f} | {w / 1e6:.1f} | {b / 1e9:.2f} | {100 * b / tot:.1f} % | {ms:.3f} | {b / ms / 1e9:.2f} |")
open(f"profiles/{tag}_hbm_bytes_by_kernel.md", "w").write("\n".join(out) + "\n")
print("\n".join(out[:30]))
