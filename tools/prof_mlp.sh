#!/bin/bash
# GPU box: k_mlp_fwd phase stamps.  Builds a -DCRD_MLP_PROF copy of the library NEXT TO the product one (never over it) and runs
# tools/prof_mlp.py with it through CRD_LIB.
set -e
cd "$(dirname "$0")/.."
OUT=camradepth_amd/libcamradepth_prof.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Wno-inline-asm -DCRD_MLP_PROF -c camradepth_amd/csrc/mlp_fused.hip -o /tmp/mlp_prof.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT $(ls camradepth_amd/csrc/build/*.o | grep -v mlp_fused.o) /tmp/mlp_prof.o
CRD_LIB=$PWD/$OUT PYTHONPATH=. python3 tools/prof_mlp.py "$@"
