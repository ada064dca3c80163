#!/bin/bash
O=gpurun_out/r6; mkdir -p $O
rm -f $O/ab_ps2.txt
for k in 0 4 8 12 14 15 0 4 8 12 14 15; do
  CRD_DEV_SWITCHES=1 CRD_PARTSUM_K=$k timeout 120 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('partsum=$k', d['ms_per_step'], d.get('ms_per_step_median'))" | tee -a $O/ab_ps2.txt
done
CRD_DEV_SWITCHES=1 CRD_PARTSUM_K=15 PYTHONPATH=. timeout 300 python tools/chain_table.py bwd 0 2000 > $O/chain_bwd_ps.log 2>/dev/null
