#!/bin/bash
O=gpurun_out/r6; mkdir -p $O
python -m pytest tests/test_gpu_gnconv.py tests/test_gpu_blocks.py -x -q > $O/t_gnconv.log 2>&1; tail -3 $O/t_gnconv.log
python -m pytest tests/test_gpu_train.py -x -q -k "shallow or bit_reproducible" > $O/t_train.log 2>&1; tail -3 $O/t_train.log
for cfg in "15 15" "0 0" "12 15" "15 15" "0 0" "12 15"; do
  set -- $cfg
  CRD_DEV_SWITCHES=1 CRD_GNB_FC1=$1 CRD_GNB_SR=$2 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fc1=$1 sr=$2', d['ms_per_step'], d.get('ms_per_step_median'))" | tee -a $O/ab_gnb2.txt
done
python bench.py --inference --batch 8 --steps 30 2>/dev/null | tail -1 | cut -c1-300
CRD_DEV_SWITCHES=1 CRD_GNB_FC1=15 CRD_GNB_SR=15 PYTHONPATH=. python tools/chain_table.py bwd 0 2000 > $O/chain_bwd_gnb.log 2>/dev/null
PYTHONPATH=. python tools/chain_table.py fwd 0 2000 > $O/chain_fwd.log 2>/dev/null
tail -n 1 $O/chain_bwd_gnb.log $O/chain_fwd.log
