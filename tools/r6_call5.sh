#!/bin/bash
O=gpurun_out/r6; mkdir -p $O
timeout 200 python -m pytest tests/test_gpu_gnconv.py -x -q -k "partsum or gn_bwd_conv" > $O/t_partsum.log 2>&1; tail -3 $O/t_partsum.log
timeout 300 python -m pytest tests/test_gpu_train.py -x -q -k "shallow or bit_reproducible or benchmark_size" > $O/t_train.log 2>&1; tail -3 $O/t_train.log
rm -f $O/ab_ps.txt
for k in 15 0 15 0; do
  CRD_DEV_SWITCHES=1 CRD_PARTSUM_K=$k timeout 120 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('partsum=$k', d['ms_per_step'], d.get('ms_per_step_median'))" | tee -a $O/ab_ps.txt
done
