"""Developer tool: print per-kernel counter averages from a rocprofv3 --pmc database. Usage: pmc_dump.py <db> [name filter]"""
import collections, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
rows = db.execute("select kernel_name, counter_name, value from counters_collection").fetchall()
agg = collections.defaultdict(lambda: [0, 0.0])
for k, c, v in rows:
    if flt in k:
        key = (k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0], c)
        agg[key][0] += 1
        agg[key][1] += v
for (k, c), (n, v) in sorted(agg.items()):
    print(f"{k:50s} {c:32s} n={n:4d} avg={v / n:16.1f}")
