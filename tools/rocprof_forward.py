"""Per-kernel totals of ONE graph-replayed eval forward (bench.py --inference) from a rocprofv3 rocpd database: forwards are
delimited by k_nchw_to_pm (the input conversion at the head of a forward); the third-last complete one is reported.
Usage: python tools/rocprof_forward.py <results.db>"""
import collections
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
ks = db.execute("select start,end,name from kernels order by start").fetchall()
packs = [s for s, e, n in ks if "k_nchw_to_pm" in n]
a, b = (packs[-4], packs[-3]) if len(packs) >= 4 else (packs[0], packs[1])
agg = collections.defaultdict(lambda: [0, 0])
first, last = None, None
for s, e, n in ks:
    if a <= s < b:
        key = n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        agg[key][0] += 1
        agg[key][1] += e - s
        first = s if first is None else first
        last = e
print(f"# one replayed eval forward (bench.py --inference --batch 8, 256x416): {sum(v[0] for v in agg.values())} launches, "
      f"kernel time {sum(v[1] for v in agg.values()) / 1e6:.3f} ms, first launch -> last end {(last - first) / 1e6:.3f} ms, "
      f"replay period {(b - a) / 1e6:.3f} ms")
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{t / 1e6:8.3f} ms {c:5d}x {t / c / 1e3:8.2f} us  {k}")
