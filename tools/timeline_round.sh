# GPU box: kernel timeline of one replayed step -> gpurun_out/timeline/step.txt
cd /tmp; export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/timeline; mkdir -p gpurun_out/timeline
rocprofv3 --kernel-trace -d gpurun_out/timeline/trace -o bench -- python3 bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/timeline/bench.log 2>&1
db=$(ls gpurun_out/timeline/trace/*.db | head -1)
python3 tools/rocprof_timeline.py "$db" gpurun_out/timeline/step.txt
python3 tools/rocprof_step.py "$db" > gpurun_out/timeline/one_step.txt
rm -rf gpurun_out/timeline/trace
wc -l gpurun_out/timeline/step.txt
