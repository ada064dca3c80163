# GPU box: kernel trace of the default bench command + one-step table; results under gpurun_out/prof_round/
cd /tmp; export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_round; mkdir -p gpurun_out/prof_round
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_round/trace -o bench -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-excess > gpurun_out/prof_round/bench.log 2>&1
tail -1 gpurun_out/prof_round/bench.log
db=$(ls gpurun_out/prof_round/trace/*.db | head -1)
python3 tools/rocprof_summary.py "$db" gpurun_out/prof_round/kernel_stats.md > /dev/null
python3 tools/rocprof_step.py "$db" > gpurun_out/prof_round/one_step.txt
head -40 gpurun_out/prof_round/one_step.txt
rm -rf gpurun_out/prof_round/trace
