# GPU box: everything profiles/r04_* holds, in one call.  Results under gpurun_out/final4/
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/final4; rm -rf $O; mkdir -p $O
( time timeout 1800 python -m pytest tests -m gpu -q -s 2>&1 ) > $O/gpu_tests.log 2>&1; tail -3 $O/gpu_tests.log
timeout 900 python bench.py > $O/bench_c2.json 2> $O/bench_c2.err; cut -c1-300 $O/bench_c2.json
timeout 600 bash tools/profile_round.sh > $O/profile_round.log 2>&1; cp gpurun_out/prof_round/kernel_stats.md gpurun_out/prof_round/one_step.txt gpurun_out/prof_round/bench.log $O/
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
rm -rf $O/fwd_trace; timeout 300 rocprofv3 --kernel-trace --stats -d $O/fwd_trace -o fwd -- python3 bench.py --inference --batch 8 --steps 20 > $O/bench_inf_b8.json 2> $O/bench_inf_b8.err
python3 tools/rocprof_forward.py $(ls $O/fwd_trace/*.db | head -1) > $O/forward_only_kernels.txt 2>&1; rm -rf $O/fwd_trace; head -5 $O/forward_only_kernels.txt
timeout 900 bash tools/pmc_round.sh > $O/pmc_round.log 2>&1; cp gpurun_out/pmc_round/traffic.json $O/pmc_traffic.json
timeout 900 bash tools/pmc_encoder.sh > $O/pmc_encoder.log 2>&1; cp gpurun_out/pmc_encoder/summary.txt $O/pmc_encoder_summary.txt
# the persistent encoder stage: stage timings against the per-launch path + per-phase stamps (a -DCRD_ENC_PROF build next to the product library)
timeout 600 bash tools/prof_enc_stage.sh 8 > $O/enc_stage_phases.txt 2> $O/enc_stage_phases.err; head -6 $O/enc_stage_phases.txt
CRD_ENC_PERSIST=1 timeout 600 python bench.py --no-cpu-baseline --no-roofline > $O/bench_c2_enc_persist.json 2> $O/bench_c2_enc_persist.err; cut -c1-200 $O/bench_c2_enc_persist.json
CRD_ENC_PERSIST=1 timeout 600 python bench.py --inference --batch 8 --steps 20 > $O/bench_inf_b8_enc_persist.json 2>> $O/bench_inf.err; cut -c1-200 $O/bench_inf_b8_enc_persist.json
timeout 600 python bench.py --variant supervised_seg --no-cpu-baseline > $O/bench_c3.json 2> $O/bench_c3.err; cut -c1-200 $O/bench_c3.json
timeout 600 python bench.py --batch 4 --height 928 --width 1600 --freeze-seg --variant supervised_seg --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_c4.json 2> $O/bench_c4.err; cut -c1-200 $O/bench_c4.json
timeout 600 python bench.py --batch 16 --no-cpu-baseline --no-roofline > $O/bench_c5_b16_bf16.json 2> $O/bench_c5.err; cut -c1-200 $O/bench_c5_b16_bf16.json
timeout 600 python bench.py --batch 16 --fp8 --no-cpu-baseline --no-roofline > $O/bench_c5_b16_fp8fwd.json 2>> $O/bench_c5.err; cut -c1-200 $O/bench_c5_b16_fp8fwd.json
timeout 600 python bench.py --inference --batch 16 --steps 20 > $O/bench_inf_b16.json 2> $O/bench_inf.err; cut -c1-200 $O/bench_inf_b16.json
timeout 600 python bench.py --inference --fp8 --batch 16 --steps 20 > $O/bench_inf_fp8_b16.json 2>> $O/bench_inf.err; cut -c1-200 $O/bench_inf_fp8_b16.json
timeout 600 python bench.py --inference --batch 1 --height 416 --width 800 --steps 50 > $O/bench_inf_b1.json 2>> $O/bench_inf.err; cut -c1-200 $O/bench_inf_b1.json
CRD_FORCE_DIST=1 timeout 600 python bench.py --no-cpu-baseline --no-roofline > $O/bench_c2_forced_dist_1rank.json 2> $O/bench_dist.err; cut -c1-300 $O/bench_c2_forced_dist_1rank.json
PYTHONPATH=. timeout 900 python tools/train_synth_checkpoint.py 3000 $O/trained_synth.pth > $O/trained_rmse.txt 2>&1; rm -f $O/trained_synth.pth; tail -4 $O/trained_rmse.txt
