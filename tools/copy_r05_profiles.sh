# gpurun_out/r05p (tools/run_r05_profiles.sh) -> profiles/r05_*: the files the documents cite.
O=gpurun_out/r05p
for f in bench_line bench_under_rocprof bench_noextras_under_rocprof train_step_bf16_bench_line train_step_bf16x3_bench_line train_step_bf16_f32act_bench_line train_step_bf16_premixed_bench_line demucs_bench_line train_step_bf16_bench_under_rocprof \
         config2_bench_under_rocprof config2_dejavu_bench_under_rocprof pmc_traffic_bf16x3 config2_sq_pass1 config2_sq_pass2 pmc_sq_pass1 pmc_sq_pass2 pmc_sq_pass3; do cp $O/$f.json profiles/r05_$f.json; done
for f in bench_kernel_stats bench_noextras_kernel_stats train_step_bf16_kernel_stats config2_kernel_stats config2_dejavu_kernel_stats; do cp $O/$f.csv profiles/r05_$f.csv; done
cp $O/conv_layers_lds_vs_direct.txt profiles/r05_conv_layers_lds_vs_direct.txt
grep -v amdgpu $O/c64_new_vs_old.txt > profiles/r05_c64_new_vs_old.txt
grep -v amdgpu $O/ws_timeline.txt > profiles/r05_ws_timeline.txt
cp $O/bench_new_vs_old.txt profiles/r05_bench_new_vs_old.txt
cp $O/pmc_sq_table.md profiles/r05_pmc_sq_table.md
grep -v amdgpu $O/small_kernels.txt > profiles/r05_small_kernels.txt
grep -v amdgpu $O/pick_stage.txt > profiles/r05_pick_stage.txt; grep -v amdgpu $O/dejavu_stages.txt > profiles/r05_dejavu_stages.txt

cp $O/train_step_launches.txt profiles/r05_train_step_launches.txt; cp $O/config2_batch_streams_lines.txt profiles/r05_config2_batch_streams_lines.txt
