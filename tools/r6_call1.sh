#!/bin/bash
# round 6, call 1: parity of crd_gn_bwd_conv + the train-step tests that exercise it, then an in-step A/B of the two fusions
O=gpurun_out/r6; mkdir -p $O
python -m pytest tests/test_gpu_gnconv.py -x -q -k "gn_bwd_conv" > $O/t_gnbwd.log 2>&1; tail -3 $O/t_gnbwd.log
python -m pytest tests/test_gpu_train.py -x -q -k "shallow or bit_reproducible or ragged" > $O/t_train.log 2>&1; tail -3 $O/t_train.log
for cfg in "15 15" "0 0" "15 0" "0 15" "12 15" "15 15" "0 0"; do
  set -- $cfg
  CRD_DEV_SWITCHES=1 CRD_GNB_FC1=$1 CRD_GNB_SR=$2 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fc1=$1 sr=$2', d['ms_per_step'], d.get('ms_per_step_median'))" | tee -a $O/ab_gnb.txt
done
