"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of an eager bench run.
Usage: python tools/pmc_traffic.py <fetch.db> <write.db> [out.json]
FETCH_SIZE / WRITE_SIZE are KiB.  Correction per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): on gfx950
FETCH_SIZE reports half the bytes of wide (16 B / lane) coalesced reads -- global_load and buffer_load...lds alike --
so it is doubled; WRITE_SIZE is taken as reported (uncalibrated in the guide)."""
import collections, json, re, sqlite3, sys


def per_kernel(path, counter):
    db = sqlite3.connect(path)
    rows = db.execute("select kernel_name, value from counters_collection where counter_name = ?", (counter,)).fetchall()
    agg = collections.defaultdict(lambda: [0, 0.0])
    for name, v in rows:
        key = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        agg[key][0] += 1
        agg[key][1] += v * 1024.0
    return agg


f, w = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(set(f) | set(w), key=lambda k: -(2 * f.get(k, [0, 0])[1] + w.get(k, [0, 0])[1])):
    nf, bf = f.get(k, [0, 0.0])
    nw, bw = w.get(k, [0, 0.0])
    n = max(nf, nw, 1)
    out[k] = {"launches": n, "fetch_bytes_per_launch": 2 * bf / n, "write_bytes_per_launch": bw / n,
              "hbm_bytes_per_launch": (2 * bf + bw) / n}
for k, v in list(out.items())[:25]:
    print(f"{v['hbm_bytes_per_launch'] / 1e6:10.2f} MB/launch  fetch {v['fetch_bytes_per_launch'] / 1e6:9.2f}  write {v['write_bytes_per_launch'] / 1e6:9.2f}  x{v['launches']:5d}  {k}")
if len(sys.argv) > 3:
    import os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from bench import csrc_sha
    json.dump({"csrc_sha": csrc_sha(), "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), eager bench.py --steps 2 --warmup 1; FETCH_SIZE doubled per the gfx950 correction",
               "kernels": out}, open(sys.argv[3], "w"), indent=1)
