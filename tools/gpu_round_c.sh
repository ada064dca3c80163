cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r02c
( time timeout 1500 python -m pytest tests/test_gpu_fp8.py -m gpu -q -x -s 2>&1 ) > gpurun_out/r02c/tests_f8.log 2>&1
grep -n "fp8 inference:\|passed\|failed" gpurun_out/r02c/tests_f8.log | cut -c1-400
for b in 8 16; do
timeout 600 python bench.py --steps 20 --inference --batch $b > gpurun_out/r02c/bench_inf_b$b.json 2> gpurun_out/r02c/bench_inf.err; cut -c1-330 gpurun_out/r02c/bench_inf_b$b.json
timeout 600 python bench.py --steps 20 --inference --fp8 --batch $b > gpurun_out/r02c/bench_inf_fp8_b$b.json 2> gpurun_out/r02c/bench_inf_fp8.err; cut -c1-330 gpurun_out/r02c/bench_inf_fp8_b$b.json; tail -3 gpurun_out/r02c/bench_inf_fp8.err
done
