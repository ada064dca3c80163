#!/bin/bash
# Builds a copy of the library with per-phase stamps in the persistent encoder stage (-DCRD_ENC_PROF) next to the product library
# (never over it) and runs tools/prof_enc_stage.py with it (CRD_LIB).  Run from the repository root on an MI355X box.
set -e
OUT=camradepth_amd/libcamradepth_prof.so
OBJ=/tmp/enc_stage_prof.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Wno-inline-asm -fno-slp-vectorize -DCRD_ENC_PROF $ENC_DEFS -c camradepth_amd/csrc/enc_stage.hip -o $OBJ
OBJS=$(ls camradepth_amd/csrc/build/*.o | grep -v enc_stage.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT $OBJS $OBJ
if [ -n "$ENC_TOOL" ]; then for i in 1 2 3 4 5 6 7 8; do CRD_LIB=$PWD/$OUT PYTHONPATH=. python $ENC_TOOL 2>/dev/null; done; exit 0; fi
CRD_LIB=$PWD/$OUT PYTHONPATH=. python tools/prof_enc_stage.py "$@"
