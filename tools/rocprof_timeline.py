"""Kernel-by-kernel timeline of ONE graph-replayed training step from a rocprofv3 rocpd database: start offset, duration, gap
to the previous kernel's end on the same stream and name.  Usage: python tools/rocprof_timeline.py <results.db> [out.txt]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)").fetchall()]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
ks = db.execute(f"select start,end,name{',' + qcol if qcol else ''} from kernels order by start").fetchall()
packs = [k[0] for k in ks if "k_nchw_to_pm" in k[2]]
upd = [k[0] for k in ks if "k_dgn_update" in k[2]]
steps = [(a, b) for a, b in zip(packs, packs[1:]) if not upd or any(a < u < b for u in upd)]      # (no optimizer: replayed inference forwards)
a, b = steps[-3] if len(steps) >= 3 else steps[-1]
last_end = {}
out = []
for k in ks:
    s, e, n = k[0], k[1], k[2]
    q = k[3] if qcol else 0
    if not (a <= s < b):
        continue
    name = n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
    last_end[q] = e
    out.append(f"{(s - a) / 1e3:10.1f} us  q{q}  dur {(e - s) / 1e3:8.2f}  gap {gap:7.2f}  {name}")
text = "\n".join(out)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(text + "\n")
else:
    print(text)
