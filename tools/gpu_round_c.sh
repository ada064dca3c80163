cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r02c
( timeout 1200 python -m pytest tests/test_gpu_train.py tests/test_gpu_model.py -m gpu -q -x 2>&1 | tail -3 )
for i in 1 2; do timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline | cut -c50-170; done
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --update-interval 3 | cut -c50-170
timeout 300 python tools/overfit_check.py 2>&1 | tail -2
