cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r02c
for d in 0 1 2 3 4 7 8; do echo "CRD_DBG=$d"; CRD_DBG=$d timeout 300 python tools/bench_gnconv.py 12 2>&1 | tail -1; CRD_DBG=$d timeout 300 python tools/bench_gnconv.py 6 2>&1 | tail -1; CRD_DBG=$d timeout 300 python tools/bench_gnconv.py 1 2>&1 | tail -1; done
