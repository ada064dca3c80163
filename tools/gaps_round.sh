cd /tmp; export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/gaps; mkdir -p gpurun_out/gaps
rocprofv3 --kernel-trace -d gpurun_out/gaps/trace -o bench -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/gaps/bench.log 2>&1
db=$(ls gpurun_out/gaps/trace/*.db | head -1)
python3 tools/rocprof_gaps.py "$db" | tee gpurun_out/gaps/gaps.txt
rm -rf gpurun_out/gaps/trace
