# gpurun_out/r06p (tools/run_r06_profiles.sh) -> profiles/r06_*: the files the documents cite.
O=gpurun_out/r06p
for f in bench_line bench_under_rocprof bench_noextras_under_rocprof bench_fp32_noextras_under_rocprof train_step_bf16_bench_line train_step_bf16_premixed_bench_line demucs_bench_line \
         train_step_bf16_bench_under_rocprof demucs_bench_under_rocprof config2_bench_under_rocprof pmc_traffic_bf16x3 pmc_sq_pass1 pmc_sq_pass2 pmc_sq_pass3; do cp $O/$f.json profiles/r06_$f.json; done
for f in bench_kernel_stats bench_noextras_kernel_stats bench_fp32_noextras_kernel_stats train_step_bf16_kernel_stats demucs_kernel_stats config2_kernel_stats; do cp $O/$f.csv profiles/r06_$f.csv; done
for f in upconv_levels upconv_levels_fp32 demucs_lstm_step small_kernels; do cp $O/$f.txt profiles/r06_$f.txt; done
cp $O/pmc_sq_table.md profiles/r06_pmc_sq_table.md
