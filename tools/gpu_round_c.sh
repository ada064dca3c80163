cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r02c
( time timeout 1500 python -m pytest tests/test_gpu_igemm.py tests/test_gpu_gnconv.py tests/test_gpu_fp8.py tests/test_gpu_ops.py -m gpu -q -x 2>&1 | tail -8 ) > gpurun_out/r02c/tests_k.log 2>&1
tail -4 gpurun_out/r02c/tests_k.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > gpurun_out/r02c/bench_c2.json 2> gpurun_out/r02c/bench_c2.err; cut -c1-200 gpurun_out/r02c/bench_c2.json
CRD_GN_CONV=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > gpurun_out/r02c/bench_c2_gn.json 2> gpurun_out/r02c/bench_c2_gn.err; cut -c1-200 gpurun_out/r02c/bench_c2_gn.json
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --inference > gpurun_out/r02c/bench_inf.json 2> gpurun_out/r02c/bench_inf.err; cut -c1-200 gpurun_out/r02c/bench_inf.json
CRD_GN_CONV=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --inference > gpurun_out/r02c/bench_inf_gn.json 2> gpurun_out/r02c/bench_inf_gn.err; cut -c1-200 gpurun_out/r02c/bench_inf_gn.json
