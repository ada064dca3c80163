#!/bin/bash
# Developer sweep: where a small encoder GEMM launch spends its time -- tools/bench_small_gemm.py under k_igemm's ablation bits
# (CRD_DBG: 8 = return at once, 16 = no K loop, 4 = no epilogue) and tile choices (CRD_IGEMM_FORCE).
# Builds a -DCRD_DEV_SWITCHES copy of the library next to the product one (never over it).
set -e
OUT=camradepth_amd/libcamradepth_dev.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Wno-inline-asm -DCRD_DEV_SWITCHES -c camradepth_amd/csrc/igemm.hip -o /tmp/igemm_dev.o
OBJS=$(ls camradepth_amd/csrc/build/*.o | grep -v /igemm.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT $OBJS /tmp/igemm_dev.o
export CRD_LIB=$PWD/$OUT PYTHONPATH=.
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Wno-inline-asm -DCRD_DEV_SWITCHES -c camradepth_amd/csrc/gngemm.hip -o /tmp/gngemm_dev.o
OBJS=$(ls camradepth_amd/csrc/build/*.o | grep -v /igemm.o | grep -v /gngemm.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT $OBJS /tmp/igemm_dev.o /tmp/gngemm_dev.o
if [ "$1" = "wide" ]; then          # the wide pointwise kernel: columns per workgroup, workgroups per CU
  for cfg in "0 2 0" "0 2 32" "0 2 0" "0 2 32"; do
    set -- $cfg
    echo "== CRD_PW_WCT2=$1 CRD_PW_OCC=$2 CRD_DBG=$3 (32: no output stores)"
    CRD_PW_WCT2=$1 CRD_PW_OCC=$2 CRD_DBG=$3 python tools/bench_small_gemm.py stats 2>/dev/null | grep -E "Cout  512|Cout 1024|Cout  640"
  done
  exit 0
fi
if [ "$1" = "tiles" ]; then          # tile choices of k_igemm (CRD_IGEMM_FORCE) per shape
  for f in 0 1 2 3 4 5 6; do
    echo "== CRD_IGEMM_FORCE=$f stats"
    CRD_IGEMM_FORCE=$f python tools/bench_small_gemm.py stats 2>/dev/null
  done
  exit 0
fi
for dbg in 0 8 16 4; do
  echo "== CRD_DBG=$dbg $1"
  CRD_DBG=$dbg python tools/bench_small_gemm.py $1 2>/dev/null
done
