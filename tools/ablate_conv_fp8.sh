# Developer tool: fp8 3x3 conv main loop with parts compiled out (see conv3x3_fp8.hip: CRD_CONV3_ABLATE).
# build here: bash tools/ablate_conv_fp8.sh build ; on the GPU box: bash tools/ablate_conv_fp8.sh
set -e
cd "$(dirname "$0")/.."
VARS="${VARS:-0 512 128 64 256 32 16 320 448 960 1008}"
if [ "$1" = "build" ]; then
  python -m camradepth_amd.build >/dev/null
  O=camradepth_amd/csrc/build
  for v in $VARS; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result -Iinclude -DCRD_CONV3_ABLATE=$v $EXTRA_DEFS -c camradepth_amd/csrc/conv3x3_fp8.hip -o /tmp/f8_a$v.o 2>/dev/null &
  done
  wait
  for v in $VARS; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o camradepth_amd/libabl_$v.so $(ls $O/*.o | grep -v /conv3x3_fp8.o) /tmp/f8_a$v.o
  done
  exit 0
fi
export PYTHONPATH=.       # (each variant is loaded through CRD_LIB: the product library is never replaced)
for v in $VARS; do
  export CRD_LIB=$PWD/camradepth_amd/libabl_$v.so
  echo "ablate=$v: $(python tools/bench_conv_fp8.py 20 | tail -1)  |  $(CIN=144 COUT=96 python tools/bench_conv_fp8.py 20 | tail -1) | $(CIN=240 COUT=64 python tools/bench_conv_fp8.py 20 | tail -1)"
done
