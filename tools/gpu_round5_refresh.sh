# GPU box: the part of profiles/r05_* that depends on the kernel sources (after a csrc change): GPU tests, the C2 bench line, the step trace,
# the PMC traffic passes (they carry the hash of csrc/), the floor table, the forward trace.  Results under gpurun_out/final5/
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/final5; mkdir -p $O
( time timeout 1800 python -m pytest tests -m gpu -q -s 2>&1 ) > $O/gpu_tests.log 2>&1; tail -3 $O/gpu_tests.log
timeout 900 python bench.py > $O/bench_c2.json 2> $O/bench_c2.err; cut -c1-300 $O/bench_c2.json
timeout 600 bash tools/profile_round.sh > $O/profile_round.log 2>&1; cp gpurun_out/prof_round/kernel_stats.md gpurun_out/prof_round/one_step.txt gpurun_out/prof_round/bench.log $O/
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
rm -rf $O/fwd_trace; timeout 300 rocprofv3 --kernel-trace --stats -d $O/fwd_trace -o fwd -- python3 bench.py --inference --batch 8 --steps 20 > $O/bench_inf_b8.json 2> $O/bench_inf_b8.err
python3 tools/rocprof_forward.py $(ls $O/fwd_trace/*.db | head -1) > $O/forward_only_kernels.txt 2>&1; rm -rf $O/fwd_trace; head -3 $O/forward_only_kernels.txt
timeout 900 bash tools/pmc_round.sh > $O/pmc_round.log 2>&1; cp gpurun_out/pmc_round/traffic.json $O/pmc_traffic.json
MS=$(python3 -c "import json; print(json.load(open('$O/bench_c2.json'))['ms_per_step'])")
PYTHONPATH=. timeout 600 python tools/floor_table.py --md $O/floor_budget.md --step-ms $MS > $O/floor_table.log 2>&1; tail -3 $O/floor_table.log
bash tools/ab_rege.sh
