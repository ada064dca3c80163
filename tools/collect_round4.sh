# build container: gpurun_out/final4 (tools/gpu_round4_final.sh) -> profiles/r04_*
cd "$(dirname "$0")/.."; O=gpurun_out/final4; P=profiles
cp $O/gpu_tests.log $P/r04_gpu_tests.log
cp $O/bench_c2.json $P/r04_bench_c2.json
cp $O/kernel_stats.md $P/r04_bench_kernel_stats.md
cp $O/one_step.txt $P/r04_one_step_kernels.txt
cp $O/forward_only_kernels.txt $P/r04_forward_only_kernels.txt
cp $O/pmc_traffic.json $P/r04_pmc_traffic.json; cp $O/pmc_traffic.json $P/pmc_traffic.json
python3 tools/pmc_encoder_md.py $O/pmc_encoder_summary.txt $P/r04_pmc_kernels.md
cp $O/enc_stage_phases.txt $P/r04_enc_stage_phases.txt
cp $O/bench_c2_enc_persist.json $P/r04_bench_c2_enc_persist.json
cp $O/bench_inf_b8_enc_persist.json $P/r04_bench_inference_b8_enc_persist.json
cp $O/bench_c3.json $P/r04_bench_c3_supervised_seg.json
cp $O/bench_c4.json $P/r04_bench_c4_928x1600_seg_frozen.json
cp $O/bench_c5_b16_bf16.json $P/r04_bench_c5_b16_bf16_train.json
cp $O/bench_c5_b16_fp8fwd.json $P/r04_bench_c5_b16_fp8fwd_train.json
cp $O/bench_inf_b16.json $P/r04_bench_inference_b16_bf16.json
cp $O/bench_inf_fp8_b16.json $P/r04_bench_inference_b16_fp8.json
cp $O/bench_inf_b1.json $P/r04_bench_inference_b1_416x800.json
cp $O/bench_inf_b8.json $P/r04_bench_inference_b8.json
cp $O/bench_c2_forced_dist_1rank.json $P/r04_bench_c2_forced_dist_1rank.json
cp $O/trained_rmse.txt $P/r04_trained_operating_point_rmse.txt
