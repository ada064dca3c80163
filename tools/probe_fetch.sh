# GPU box: FETCH_SIZE calibration on the conv's 64-byte-row LDS-DMA pattern -> gpurun_out/probe_fetch/
cd /tmp; export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/probe_fetch; mkdir -p gpurun_out/probe_fetch
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probe_fetch.hip -o /tmp/probe_fetch || exit 1
/tmp/probe_fetch > gpurun_out/probe_fetch/timing.txt 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/probe_fetch/pmc -o p -- /tmp/probe_fetch > gpurun_out/probe_fetch/pmc_run.txt 2>&1
db=$(ls gpurun_out/probe_fetch/pmc/*.db | head -1)
python3 - "$db" > gpurun_out/probe_fetch/fetch.txt <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(counters_collection)").fetchall()]
order = "dispatch_id" if "dispatch_id" in cols else ("id" if "id" in cols else cols[0])
rows = db.execute(f"select kernel_name, value from counters_collection where counter_name = 'FETCH_SIZE' order by {order}").fetchall()
labels = ["rows64 chunk0", "rows64 chunk1", "rows64 chunk0 (b2b)", "rows64 chunk1 (b2b, right behind chunk0)", "rows128 chunk0", "contig 54.5 MB", "contig 109 MB", "contig 518 MB"]
useful = [54.5, 54.5, 54.5, 54.5, 109.05, 54.5, 109.05, 518.0]
ks = [(n, v) for n, v in rows if "k_rows" in n or "k_contig" in n]
print("FETCH_SIZE as reported (KiB -> MB), per launch, in program order; 'x2' = the guide's gfx950 correction for wide coalesced reads")
for (n, v), lab, u in zip(ks, labels, useful):
    mb = v * 1024 / 1e6
    print(f"{lab:44s} reported {mb:8.1f} MB   x2 {2 * mb:8.1f} MB   useful {u:7.1f} MB   reported/useful {mb / u:5.2f}")
PY
cat gpurun_out/probe_fetch/timing.txt gpurun_out/probe_fetch/fetch.txt
rm -rf gpurun_out/probe_fetch/pmc
