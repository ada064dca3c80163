# Developer tool: builds camradepth_amd/libprof.so = the library with conv3x3.hip compiled -DCRD_CONV3_PROF, and runs
# tools/bench_conv.py against it (on the GPU box: bash tools/build_prof.sh run <bench_conv args>)
set -e
cd "$(dirname "$0")/.."
if [ "$1" = "run" ]; then          # (the profiling library is loaded through CRD_LIB: the product library is never replaced)
  shift
  CRD_LIB=$PWD/camradepth_amd/libprof.so PYTHONPATH=. python tools/bench_conv.py "$@" || true
  exit 0
fi
python -m camradepth_amd.build >/dev/null
O=camradepth_amd/csrc/build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result -Iinclude ${PROF_DEFS:--DCRD_CONV3_PROF} -c camradepth_amd/csrc/conv3x3.hip -o /tmp/conv3x3_prof.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o camradepth_amd/libprof.so $(ls $O/*.o | grep -v conv3x3.o) /tmp/conv3x3_prof.o
