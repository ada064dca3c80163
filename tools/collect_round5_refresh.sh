# build container: the refreshed part of gpurun_out/final5 -> profiles/r05_*
cd "$(dirname "$0")/.."; O=gpurun_out/final5; P=profiles
cp $O/gpu_tests.log $P/r05_gpu_tests.log
cp $O/bench_c2.json $P/r05_bench_c2.json
cp $O/kernel_stats.md $P/r05_bench_kernel_stats.md
cp $O/one_step.txt $P/r05_one_step_kernels.txt
cp $O/forward_only_kernels.txt $P/r05_forward_only_kernels.txt
cp $O/bench_inf_b8.json $P/r05_bench_inference_b8.json
cp $O/pmc_traffic.json $P/r05_pmc_traffic.json; cp $O/pmc_traffic.json $P/pmc_traffic.json
cp $O/floor_budget.md $P/r05_floor_budget.md
cp gpurun_out/r5/ab_rege.txt $P/r05_ab_register_epilogue.txt
