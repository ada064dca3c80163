cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r02a
( time timeout 3000 python -m pytest tests -m gpu -q 2>&1 | tail -30 ) > gpurun_out/r02a/tests.log 2>&1
tail -6 gpurun_out/r02a/tests.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r02a/bench_c2.json 2> gpurun_out/r02a/bench_c2.err; cut -c1-260 gpurun_out/r02a/bench_c2.json
