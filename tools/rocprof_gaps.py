"""GPU idle time inside ONE graph-replayed training step from a rocprofv3 rocpd database: the union of all kernel intervals
(both streams) against the step's wall time, the distribution of the gaps between consecutive busy intervals, and the kernels
that precede the longest gaps.  Usage: python tools/rocprof_gaps.py <results.db>"""
import collections, sqlite3, sys

db = sqlite3.connect(sys.argv[1])
ks = db.execute("select start,end,name from kernels order by start").fetchall()
packs = [s for s, e, n in ks if "k_nchw_to_pm" in n]
upd = [s for s, e, n in ks if "k_dgn_update" in n]
steps = [(a, b) for a, b in zip(packs, packs[1:]) if any(a < u < b for u in upd)]
a, b = steps[-3] if len(steps) >= 3 else steps[-1]
inside = [(s, e, n) for s, e, n in ks if a <= s < b]
busy, gaps, cur_s, cur_e, last_name = 0, [], None, None, None
for s, e, n in inside:
    if cur_e is None:
        cur_s, cur_e, last_name = s, e, n
        continue
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, last_name, n))
        cur_s, cur_e, last_name = s, e, n
    elif e > cur_e:
        cur_e, last_name = e, n
busy += cur_e - cur_s
wall = b - a
short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:44]
print(f"step wall {wall / 1e6:.3f} ms, {len(inside)} kernels, sum of kernel durations {sum(e - s for s, e, _ in inside) / 1e6:.3f} ms")
print(f"GPU busy (union of both streams) {busy / 1e6:.3f} ms, idle {(wall - busy) / 1e6:.3f} ms in {len(gaps)} gaps")
g = sorted(x[0] for x in gaps)
if g:
    q = lambda f: g[min(len(g) - 1, int(f * len(g)))] / 1e3
    print(f"gap median {q(0.5):.2f} us, 90 % {q(0.9):.2f} us, max {g[-1] / 1e3:.2f} us; gaps > 3 us: {sum(1 for x in g if x > 3000)} totalling {sum(x for x in g if x > 3000) / 1e6:.3f} ms")
by = collections.defaultdict(lambda: [0, 0])
for d, before, after in gaps:
    by[short(before)][0] += 1; by[short(before)][1] += d
print("idle time by the kernel that ran before the gap:")
for k, (c, t) in sorted(by.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f"  {t / 1e6:7.3f} ms {c:5d}x {t / c / 1e3:6.2f} us  after {k}")
