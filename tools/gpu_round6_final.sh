# GPU box: everything profiles/r06_* holds, in one call (~20 GPU-minutes; every step under its own timeout).  Results under gpurun_out/final6/
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/final6; rm -rf $O; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -q -s 2>&1 ) > $O/gpu_tests.log 2>&1; tail -3 $O/gpu_tests.log
timeout 900 python bench.py > $O/bench_c2.json 2> $O/bench_c2.err; cut -c1-300 $O/bench_c2.json
timeout 600 bash tools/profile_round.sh > $O/profile_round.log 2>&1; cp gpurun_out/prof_round/kernel_stats.md gpurun_out/prof_round/one_step.txt gpurun_out/prof_round/bench.log $O/
cp $O/one_step.txt profiles/r06_one_step_kernels.txt     # the second bench's frac_committed_trace reads this table
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
rm -rf $O/fwd_trace; timeout 300 rocprofv3 --kernel-trace --stats -d $O/fwd_trace -o fwd -- python3 bench.py --inference --batch 8 --steps 20 > $O/bench_inf_b8.json 2> $O/bench_inf_b8.err
python3 tools/rocprof_forward.py $(ls $O/fwd_trace/*.db | head -1) > $O/forward_only_kernels.txt 2>&1; rm -rf $O/fwd_trace; head -3 $O/forward_only_kernels.txt
timeout 900 bash tools/pmc_round.sh > $O/pmc_round.log 2>&1; cp gpurun_out/pmc_round/traffic.json $O/pmc_traffic.json
# a second default run AFTER the PMC passes: its roofline.traffic comes from the counters of these very sources
cp $O/pmc_traffic.json profiles/pmc_traffic.json
timeout 900 python bench.py > $O/bench_c2_with_traffic.json 2> $O/bench_c2b.err; cut -c1-200 $O/bench_c2_with_traffic.json
MS=$(python3 -c "import json; print(json.load(open('$O/bench_c2.json'))['ms_per_step'])")
PYTHONPATH=. timeout 600 python tools/floor_table.py --md $O/floor_budget.md --step-ms $MS > $O/floor_table.log 2>&1; tail -3 $O/floor_table.log
timeout 600 python bench.py --variant supervised_seg --no-cpu-baseline > $O/bench_c3.json 2> $O/bench_c3.err; cut -c1-200 $O/bench_c3.json
timeout 600 python bench.py --batch 4 --height 928 --width 1600 --freeze-seg --variant supervised_seg --steps 10 --warmup 3 --no-cpu-baseline --no-excess > $O/bench_c4.json 2> $O/bench_c4.err; cut -c1-200 $O/bench_c4.json
for rep in 1 2; do
timeout 600 python bench.py --batch 16 --no-cpu-baseline --no-roofline --no-excess > $O/bench_c5_b16_bf16_$rep.json 2> $O/bench_c5.err; cut -c1-160 $O/bench_c5_b16_bf16_$rep.json
timeout 600 python bench.py --batch 16 --fp8 --no-cpu-baseline --no-roofline --no-excess > $O/bench_c5_b16_fp8fwd_$rep.json 2>> $O/bench_c5.err; cut -c1-160 $O/bench_c5_b16_fp8fwd_$rep.json
timeout 600 python bench.py --batch 16 --fp8 --fp8-grad --no-cpu-baseline --no-roofline --no-excess > $O/bench_c5_b16_fp8fwd_dgrad_$rep.json 2>> $O/bench_c5.err; cut -c1-160 $O/bench_c5_b16_fp8fwd_dgrad_$rep.json
done
timeout 600 python bench.py --inference --batch 16 --steps 20 > $O/bench_inf_b16.json 2> $O/bench_inf.err; cut -c1-200 $O/bench_inf_b16.json
timeout 600 python bench.py --inference --fp8 --batch 16 --steps 20 > $O/bench_inf_fp8_b16.json 2>> $O/bench_inf.err; cut -c1-200 $O/bench_inf_fp8_b16.json
timeout 600 python bench.py --inference --batch 1 --height 416 --width 800 --steps 50 > $O/bench_inf_b1.json 2>> $O/bench_inf.err; cut -c1-200 $O/bench_inf_b1.json
CRD_FORCE_DIST=1 timeout 600 python bench.py --no-cpu-baseline --no-roofline --no-excess > $O/bench_c2_forced_dist_1rank.json 2> $O/bench_dist.err; cut -c1-300 $O/bench_c2_forced_dist_1rank.json
# config 5 per kernel: the backward chain of the decoder's two fp8 stages, bf16 against e4m3 data gradients (B = 16)
CRD_CHAIN_BATCH=16 PYTHONPATH=. timeout 600 python tools/chain_table.py bwd 0 40 > $O/chain_c5_bf16.txt 2>&1
CRD_CHAIN_BATCH=16 CRD_CHAIN_FP8=grad PYTHONPATH=. timeout 600 python tools/chain_table.py bwd 0 48 > $O/chain_c5_fp8grad.txt 2>&1; grep -E "fp8|dgrad" $O/chain_c5_fp8grad.txt | head -12
