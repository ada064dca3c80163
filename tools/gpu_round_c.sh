cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r02c
( timeout 600 python -m pytest tests/test_gpu_fp8.py -m gpu -q -k quantisation 2>&1 | grep -E "Error|error|^E" | head -20 ) 2>&1
