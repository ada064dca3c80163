#!/bin/bash
# kernel trace of the default bench command + one-step table -> gpurun_out/r6/one_step.txt
cd /tmp; export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6; mkdir -p $O; rm -rf $O/trace
timeout 400 rocprofv3 --kernel-trace --stats -d $O/trace -o bench -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $O/bench_prof.log 2>&1
tail -1 $O/bench_prof.log | cut -c1-200
db=$(ls $O/trace/*.db | head -1)
python3 tools/rocprof_step.py "$db" > $O/one_step.txt
rm -rf $O/trace
head -5 $O/one_step.txt
