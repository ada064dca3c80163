#!/bin/bash
# Developer sweep of the streaming 3x3 weight gradient (k_wgrad3x3) on the 256 x 416 level: tile shapes per wave, workgroup budgets.
# Builds a -DCRD_DEV_SWITCHES copy of the library next to the product one (never over it).
set -e
OUT=camradepth_amd/libcamradepth_dev.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Wno-inline-asm -DCRD_DEV_SWITCHES -c camradepth_amd/csrc/wgrad3x3.hip -o /tmp/wgrad3x3_dev.o
OBJS=$(ls camradepth_amd/csrc/build/*.o | grep -v wgrad3x3.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT $OBJS /tmp/wgrad3x3_dev.o
export CRD_LIB=$PWD/$OUT PYTHONPATH=.
for shape in "304 128" "360 128" "128 128"; do
  set -- $shape
  for budget in 0 160; do
    for t4 in 0 1; do
      CIN=$1 COUT=$2 BUDGET=$budget CRD_W3_T22=$((1 - t4)) python tools/bench_wgrad.py 2>/dev/null | sed "s/^/4x1-tiles $t4 /"
    done
  done
done
