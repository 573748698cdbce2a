# gpurun_out/r04p (tools/run_r04_profiles.sh) -> profiles/r04_*: the files the documents cite.  The SQ / TA passes of the conv family
# (r04_pmc_sq_pass{1,2,3}.json, tabulated by hand in r04_pmc_sq.md) are NOT overwritten: see profiles/README.md.
O=gpurun_out/r04p
for f in bench_line bench_under_rocprof train_step_bf16_bench_line train_step_bf16x3_bench_line demucs_bench_line train_step_bf16_bench_under_rocprof \
         config2_bench_under_rocprof config2_dejavu_bench_under_rocprof pmc_traffic_bf16x3 config2_sq_pass1 config2_sq_pass2; do cp $O/$f.json profiles/r04_$f.json; done
for f in bench_kernel_stats train_step_bf16_kernel_stats config2_kernel_stats config2_dejavu_kernel_stats; do cp $O/$f.csv profiles/r04_$f.csv; done
cp $O/conv_layers_lds_vs_direct.txt profiles/r04_conv_layers_lds_vs_direct.txt; cp $O/c64_layers.txt profiles/r04_c64_layers.txt
grep -v amdgpu $O/small_kernels.txt > profiles/r04_small_kernels.txt
grep -v amdgpu $O/pick_stage.txt > profiles/r04_pick_stage.txt; grep -v amdgpu $O/dejavu_stages.txt > profiles/r04_dejavu_stages.txt
