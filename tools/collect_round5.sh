# build container: gpurun_out/final5 (tools/gpu_round5_final.sh) -> profiles/r05_*
cd "$(dirname "$0")/.."; O=gpurun_out/final5; P=profiles
cp $O/gpu_tests.log $P/r05_gpu_tests.log
cp $O/bench_c2.json $P/r05_bench_c2.json
cp $O/kernel_stats.md $P/r05_bench_kernel_stats.md
cp $O/one_step.txt $P/r05_one_step_kernels.txt
cp $O/forward_only_kernels.txt $P/r05_forward_only_kernels.txt
cp $O/pmc_traffic.json $P/r05_pmc_traffic.json; cp $O/pmc_traffic.json $P/pmc_traffic.json
cp $O/floor_budget.md $P/r05_floor_budget.md
cp $O/bench_narrow.txt $P/r05_narrow_pointwise_microbench.txt
cp $O/bench_c3.json $P/r05_bench_c3_supervised_seg.json
cp $O/bench_c4.json $P/r05_bench_c4_928x1600_seg_frozen.json
cp $O/bench_c5_b16_bf16.json $P/r05_bench_c5_b16_bf16_train.json
cp $O/bench_c5_b16_fp8fwd.json $P/r05_bench_c5_b16_fp8fwd_train.json
cp $O/bench_c5_b16_fp8fwd_dgrad.json $P/r05_bench_c5_b16_fp8fwd_dgrad_train.json
cp $O/bench_inf_b16.json $P/r05_bench_inference_b16_bf16.json
cp $O/bench_inf_fp8_b16.json $P/r05_bench_inference_b16_fp8.json
cp $O/bench_inf_b1.json $P/r05_bench_inference_b1_416x800.json
cp $O/bench_inf_b8.json $P/r05_bench_inference_b8.json
cp $O/bench_c2_forced_dist_1rank.json $P/r05_bench_c2_forced_dist_1rank.json
cp $O/chain_c5_bf16.txt $P/r05_c5_decoder_backward_chain_bf16.txt
cp $O/chain_c5_fp8grad.txt $P/r05_c5_decoder_backward_chain_fp8grad.txt
cp $O/trained_rmse.txt $P/r05_trained_operating_point_rmse.txt
