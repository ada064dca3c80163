cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r02a
( time timeout 3000 python -m pytest tests -m gpu -q -s 2>&1 | tail -150 ) > gpurun_out/r02a/tests.log 2>&1
timeout 900 python bench.py --variant supervised_seg --freeze-seg --height 928 --width 1600 --batch 4 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r02a/bench_c4.json 2> gpurun_out/r02a/bench_c4.err
tail -5 gpurun_out/r02a/tests.log; cat gpurun_out/r02a/bench_c4.json | cut -c1-600; tail -3 gpurun_out/r02a/bench_c4.err
