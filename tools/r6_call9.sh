#!/bin/bash
O=gpurun_out/r6; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_fp8.py -x -q -s -k "e4m3_variant" > $O/t_attn_fp8.log 2>&1; grep "e4m3 scores\|passed\|failed" $O/t_attn_fp8.log | cut -c1-300
for b in 8 16; do PYTHONPATH=. timeout 200 python tools/bench_attn_fp8.py $b 2>/dev/null | tee -a $O/attn_fp8_bench.txt; done
timeout 900 python -m pytest tests/test_gpu_trained.py -x -q -s > $O/t_trained.log 2>&1; tail -3 $O/t_trained.log; grep -i "rmse" $O/t_trained.log | cut -c1-300 | head
