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
