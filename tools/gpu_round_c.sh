cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r02c
( time timeout 1500 python -m pytest tests/test_gpu_fp8.py -m gpu -q -x -s 2>&1 ) > gpurun_out/r02c/tests_f8.log 2>&1
grep -n "fp8 inference:\|fp8 forward in\|passed\|failed\|Error\|^E " gpurun_out/r02c/tests_f8.log | cut -c1-400
