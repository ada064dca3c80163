python -m pytest tests/test_gpu_igemm.py tests/test_gpu_model.py tests/test_gpu_data.py tests/test_gpu_fp8.py -m gpu -q -x 2>&1 | tail -3
python bench.py --inference --batch 1 --height 416 --width 800 --steps 50 2>/dev/null | cut -c1-150
python bench.py --inference --batch 16 --steps 20 2>/dev/null | cut -c1-150
python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | cut -c1-150
