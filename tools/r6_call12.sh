#!/bin/bash
O=gpurun_out/r6; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_gnconv.py -x -q > $O/t_gnconv.log 2>&1; tail -3 $O/t_gnconv.log
timeout 400 python -m pytest tests/test_gpu_train.py tests/test_gpu_blocks.py -x -q -k "shallow or bit_reproducible or every_block or benchmark_size" > $O/t_train.log 2>&1; tail -3 $O/t_train.log
timeout 900 python -m pytest tests/test_gpu_fp8.py -x -q -k "oracle or graph_step or scale" > $O/t_fp8b.log 2>&1; tail -3 $O/t_fp8b.log
rm -f $O/ab_qsr.txt
for k in 6 0 4 6 0 4; do
  CRD_DEV_SWITCHES=1 CRD_QSR_GROUP=$k timeout 120 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-roofline --no-excess 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('qsr=$k', d['ms_per_step'], d.get('ms_per_step_median'))" | tee -a $O/ab_qsr.txt
done
for k in 6 0; do CRD_DEV_SWITCHES=1 CRD_QSR_GROUP=$k timeout 120 python bench.py --inference --batch 8 --steps 30 2>/dev/null | tail -1 | cut -c1-170 | tee -a $O/ab_qsr.txt; done
