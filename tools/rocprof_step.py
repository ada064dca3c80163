"""Per-kernel totals of ONE graph-replayed training step from a rocprofv3 rocpd database (steps delimited by k_nchw_to_pm, the input conversion at the head of a forward).
Usage: python tools/rocprof_step.py <results.db> [<other.db>]   (two databases: side-by-side + fixed-cost estimate 2*t1-t2)"""
import collections, sqlite3, sys


def step_table(path):
    """A training step = the interval between two consecutive k_nchw_to_pm launches (the head of a forward) that
    contains an optimizer update; the third-last such interval of the trace is reported."""
    db = sqlite3.connect(path)
    ks = db.execute("select start,end,name from kernels order by start").fetchall()
    packs = [s for s, e, n in ks if "k_nchw_to_pm" in n]
    upd = [s for s, e, n in ks if "k_dgn_update" in n]
    steps = [(a, b) for a, b in zip(packs, packs[1:]) if any(a < u < b for u in upd)]
    a, b = steps[-3] if len(steps) >= 3 else steps[-1]
    agg = collections.defaultdict(lambda: [0, 0])
    for s, e, n in ks:
        if a <= s < b:
            key = n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            agg[key][0] += 1; agg[key][1] += e - s
    return agg, (b - a) / 1e6


t1, w1 = step_table(sys.argv[1])
if len(sys.argv) > 2:
    t2, w2 = step_table(sys.argv[2])
    print(f"step wall {w1:.2f} / {w2:.2f} ms")
    rows = []
    for k in set(t1) | set(t2):
        a, b = t1.get(k, [0, 0]), t2.get(k, [0, 0])
        rows.append((2 * a[1] - b[1], k, a, b))
    print(f"{'fixed ms':>9} {'t1 ms':>8} {'t2 ms':>8} {'n':>5}  kernel")
    for fx, k, a, b in sorted(rows, reverse=True):
        print(f"{fx / 1e6:9.3f} {a[1] / 1e6:8.3f} {b[1] / 1e6:8.3f} {b[0]:5d}  {k}")
    print(f"sum fixed {sum(r[0] for r in rows) / 1e6:.2f} ms")
else:
    print(f"step wall {w1:.2f} ms")
    for k, (c, t) in sorted(t1.items(), key=lambda kv: -kv[1][1]):
        print(f"{t / 1e6:8.3f} ms {c:5d}x {t / c / 1e3:8.2f} us  {k}")
