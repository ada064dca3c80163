#!/bin/bash
# every step under its own timeout: a faulting kernel must not hold the box until gpurun's limit
O=gpurun_out/r6; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_train.py -x -q -k "shallow or bit_reproducible or benchmark_size or config3" > $O/t_train.log 2>&1; tail -3 $O/t_train.log
timeout 400 python -m pytest tests/test_gpu_model.py -x -q > $O/t_model.log 2>&1; tail -3 $O/t_model.log
rm -f $O/ab_kcat.txt
for k in 1 0 1 0; do
  CRD_DEV_SWITCHES=1 CRD_KCAT=$k timeout 120 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('kcat=$k', d['ms_per_step'], d.get('ms_per_step_median'))" | tee -a $O/ab_kcat.txt
done
CRD_DEV_SWITCHES=1 CRD_KCAT=1 PYTHONPATH=. timeout 200 python tools/chain_table.py bwd 0 140 > $O/chain_dec_kcat.log 2>/dev/null
CRD_DEV_SWITCHES=1 CRD_KCAT=0 PYTHONPATH=. timeout 200 python tools/chain_table.py bwd 0 140 > $O/chain_dec_base.log 2>/dev/null
tail -n 1 $O/chain_dec_kcat.log $O/chain_dec_base.log
