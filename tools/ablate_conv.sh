# Developer tool: timing of the 3x3 conv main loop with parts compiled out.  bits: 256 no DMA, 16 no vmcnt wait,
# 32 no barrier, 64 no fragment reads, 128 no MFMA.   build: bash tools/ablate_conv.sh build ; GPU box: bash tools/ablate_conv.sh
set -e
cd "$(dirname "$0")/.."
VARS="${VARS:-0 256 16 32 48 64 128 304 368 496}"
FILE="${FILE:-conv3x3}"          # conv3x3 (two workgroups per CU) or conv3x3p (persistent, one wave per SIMD)
if [ "$1" = "build" ]; then
  python -m camradepth_amd.build >/dev/null
  O=camradepth_amd/csrc/build
  for v in $VARS; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result -Iinclude -DCRD_CONV3_ABLATE=$v $EXTRA_DEFS -c camradepth_amd/csrc/$FILE.hip -o /tmp/conv3x3_a$v.o 2>/dev/null &
  done
  wait
  for v in $VARS; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o camradepth_amd/libabl_$v.so $(ls $O/*.o | grep -v /$FILE.o) /tmp/conv3x3_a$v.o
  done
  exit 0
fi
export PYTHONPATH=.       # (each variant is loaded through CRD_LIB: the product library is never replaced)
for v in $VARS; do
  export CRD_LIB=$PWD/camradepth_amd/libabl_$v.so
  echo "ablate=$v: $(python tools/bench_conv.py 0 20 | tail -1)  |  $(CIN=144 COUT=96 python tools/bench_conv.py 0 20 | tail -1) | $(CIN=240 COUT=64 python tools/bench_conv.py 0 20 | tail -1)"
done
