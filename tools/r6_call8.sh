#!/bin/bash
O=gpurun_out/r6; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_fp8.py -x -q -s > $O/t_fp8.log 2>&1; tail -4 $O/t_fp8.log
rm -f $O/c5_b16.txt
for rep in 1 2; do
for mode in "" "--fp8" "--fp8 --fp8-grad"; do
  timeout 200 python bench.py --batch 16 --steps 40 --warmup 10 --no-cpu-baseline --no-roofline --no-excess $mode 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B16 [$mode]', d['value'], d['ms_per_step'])" | tee -a $O/c5_b16.txt
done; done
