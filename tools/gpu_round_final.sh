# GPU box: everything profiles/ holds for a round.  Results under gpurun_out/final/
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/final; rm -rf $O; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -q -x -s 2>&1 ) > $O/gpu_tests.log 2>&1; tail -3 $O/gpu_tests.log
timeout 900 bash tools/pmc_kernels.sh > /dev/null 2>&1; cp gpurun_out/pmc_kernels/summary.txt $O/pmc_kernels.txt
timeout 900 bash tools/pmc_round.sh > $O/pmc_round.log 2>&1; cp gpurun_out/pmc_round/traffic.json $O/pmc_traffic.json; cp $O/pmc_traffic.json profiles/pmc_traffic.json
timeout 600 bash tools/profile_round.sh > $O/profile_round.log 2>&1; cp gpurun_out/prof_round/kernel_stats.md gpurun_out/prof_round/one_step.txt gpurun_out/prof_round/bench.log $O/
timeout 900 python bench.py > $O/bench_c2.json 2> $O/bench_c2.err; cut -c1-400 $O/bench_c2.json
timeout 600 python bench.py --variant supervised_seg --no-cpu-baseline > $O/bench_c3.json 2> $O/bench_c3.err; cut -c1-200 $O/bench_c3.json
timeout 600 python bench.py --batch 4 --height 928 --width 1600 --freeze-seg --variant supervised_seg --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_c4.json 2> $O/bench_c4.err; cut -c1-200 $O/bench_c4.json
timeout 600 python bench.py --inference --batch 16 --steps 20 > $O/bench_inf_b16.json 2> $O/bench_inf.err; cut -c1-200 $O/bench_inf_b16.json
timeout 600 python bench.py --inference --fp8 --batch 16 --steps 20 > $O/bench_inf_fp8_b16.json 2>> $O/bench_inf.err; cut -c1-200 $O/bench_inf_fp8_b16.json
timeout 600 python bench.py --inference --batch 1 --height 416 --width 800 --steps 50 > $O/bench_inf_b1.json 2>> $O/bench_inf.err; cut -c1-200 $O/bench_inf_b1.json
