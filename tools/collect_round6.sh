# build container: gpurun_out/final6 (tools/gpu_round6_final.sh) -> profiles/r06_*, then the generated README block
cd "$(dirname "$0")/.."; O=gpurun_out/final6; P=profiles
cp $O/gpu_tests.log $P/r06_gpu_tests.log
cp $O/bench_c2.json $P/r06_bench_c2.json
cp $O/bench_c2_with_traffic.json $P/r06_bench_c2_with_traffic.json
cp $O/kernel_stats.md $P/r06_bench_kernel_stats.md
cp $O/one_step.txt $P/r06_one_step_kernels.txt
cp $O/forward_only_kernels.txt $P/r06_forward_only_kernels.txt
cp $O/pmc_traffic.json $P/r06_pmc_traffic.json; cp $O/pmc_traffic.json $P/pmc_traffic.json
cp $O/floor_budget.md $P/r06_floor_budget.md
cp $O/bench_c3.json $P/r06_bench_c3_supervised_seg.json
cp $O/bench_c4.json $P/r06_bench_c4_928x1600_seg_frozen.json
for r in 1 2; do
cp $O/bench_c5_b16_bf16_$r.json $P/r06_bench_c5_b16_bf16_train_$r.json
cp $O/bench_c5_b16_fp8fwd_$r.json $P/r06_bench_c5_b16_fp8fwd_train_$r.json
cp $O/bench_c5_b16_fp8fwd_dgrad_$r.json $P/r06_bench_c5_b16_fp8fwd_dgrad_train_$r.json
done
cp $O/bench_inf_b16.json $P/r06_bench_inference_b16_bf16.json
cp $O/bench_inf_fp8_b16.json $P/r06_bench_inference_b16_fp8.json
cp $O/bench_inf_b1.json $P/r06_bench_inference_b1_416x800.json
cp $O/bench_inf_b8.json $P/r06_bench_inference_b8.json
cp $O/bench_c2_forced_dist_1rank.json $P/r06_bench_c2_forced_dist_1rank.json
cp $O/chain_c5_bf16.txt $P/r06_c5_decoder_backward_chain_bf16.txt
cp $O/chain_c5_fp8grad.txt $P/r06_c5_decoder_backward_chain_fp8grad.txt
python3 tools/readme_parity.py $P/r06_gpu_tests.log
