#!/bin/bash
# A/B in one GPU call: which encoder stages take the fused GroupNorm-backward + fc1 data gradient (CRD_GNB_FC1 stage mask)
O=gpurun_out/r6; mkdir -p $O; : > $O/ab_fc1mask.txt
for rep in 1 2; do
  for m in 15 7 3 1; do
    v=$(CRD_DEV_SWITCHES=1 CRD_GNB_FC1=$m timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-excess 2>/dev/null | tail -1 | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
    echo "fc1=$m $v" | tee -a $O/ab_fc1mask.txt
  done
done
