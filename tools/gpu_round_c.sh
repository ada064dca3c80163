cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r02c
( timeout 900 python -m pytest tests/test_gpu_gnconv.py tests/test_gpu_model.py -m gpu -q -x 2>&1 | tail -3 )
timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline | cut -c1-1500
