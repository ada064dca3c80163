#!/bin/bash
# One GPU call: the engine's developer switches against the default, two passes (bench.py --steps 20, ms per step)
O=gpurun_out/r6; mkdir -p $O; : > $O/sweep.txt
run() { v=$(env CRD_DEV_SWITCHES=1 "$@" timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-excess 2>/dev/null | tail -1 | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])" 2>/dev/null); echo "$* -> $v" | tee -a $O/sweep.txt; }
for rep in 1 2; do
  run CRD_NOP=1
  for m in 6 4 3 0; do run CRD_GNB_SR=$m; done
  for w in 96 128 192 224; do run CRD_W3_LATE_WGS=$w; done
  run CRD_NOP=1
  run CRD_MLP_FUSED=0
  run CRD_MLP_FUSED_MAXPIX=512
  run CRD_FUSE_GN_RED=1
  run CRD_GN_CONV=1
  run CRD_QSR_GROUP=0
  run CRD_QSR_GROUP=12
  run CRD_NO_FUSE_NORM2_APPLY=1
  run CRD_NO_FUSE_BLOCK_RED=1
  run CRD_NOP=1
done
