"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / average / share, as the `--stats` view.
Usage: python tools/rocprof_summary.py <results.db> [out.md]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, count(*), sum(end - start), avg(end - start) from kernels group by name order by sum(end - start) desc").fetchall()
tot_all = sum(r[2] for r in rows)
rows = [(r[0], r[1], r[2], r[3], 100.0 * r[2] / tot_all) for r in rows]
span = db.execute("select min(start), max(end) from kernels").fetchone()
lines = ["| kernel | calls | total ms | avg us | % |", "|---|---|---|---|---|"]
for name, calls, tot, avg, pct in rows:
    n = name if len(name) < 110 else name[:107] + "..."
    lines.append(f"| `{n}` | {calls} | {tot / 1e6:.3f} | {avg / 1e3:.2f} | {pct:.2f} |")
lines.append("")
lines.append(f"kernel time total {sum(r[2] for r in rows) / 1e6:.2f} ms over a {(span[1] - span[0]) / 1e6:.2f} ms trace window, "
             f"{sum(r[1] for r in rows)} dispatches")
text = "\n".join(lines)
print(text)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(text + "\n")

# Idle analysis over the tail of the trace (the timed graph replays): union of busy intervals vs wall window.
W = 150e6
ks = db.execute("select start, end, name from kernels where start > ? order by start", (span[1] - W,)).fetchall()
busy = 0; cur_s, cur_e = ks[0][0], ks[0][1]; gaps = []
for s, e, _ in ks[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; gaps.append(s - cur_e); cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
wall = ks[-1][1] - ks[0][0]
gaps.sort()
tail = (f"last {wall / 1e6:.1f} ms of the trace: {len(ks)} dispatches, GPU busy {busy / 1e6:.1f} ms ({100 * busy / wall:.1f} %), "
        f"{len(gaps)} gaps, median gap {gaps[len(gaps) // 2] / 1e3:.2f} us, total gap {sum(gaps) / 1e6:.2f} ms")
print(tail)
if len(sys.argv) > 2:
    open(sys.argv[2], "a").write(tail + "\n")
