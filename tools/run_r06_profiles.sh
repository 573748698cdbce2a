# Round-6 profile set of the FINAL binary (the decoder levels folded: conv_up_kernel in the conv family): the driver's bench line, rocprofv3 kernel
# stats of the same command, PMC traffic + SQ counters of the conv family, per-level fused-vs-two-launch pairs, train / demucs lines, config 2 stats.
# Outputs under gpurun_out/r06p; tools/copy_r06_profiles.sh copies what is judged into profiles/r06_*.
export TMPDIR=/tmp
O=gpurun_out/r06p; mkdir -p $O
FAM=conv_mfma_kernel,convT_mfma_kernel,conv_wd16_kernel,conv_ws64_kernel,conv_up_kernel
timeout -k 10 700 python bench.py --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench.err
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o p -- python3 bench.py --cpu-seconds 0 --no-configs > $O/bench_under_rocprof.json 2>> $O/bench.err
cp $(find $O/kt -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv; rm -rf $O/kt
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt0 -o p -- python3 bench.py --cpu-seconds 0 --no-configs --no-extras --steps 5 --warmup 1 > $O/bench_noextras_under_rocprof.json 2>> $O/bench.err
cp $(find $O/kt0 -name "*kernel_stats.csv" | head -1) $O/bench_noextras_kernel_stats.csv; rm -rf $O/kt0
# the same without extras at the reference's arithmetic (exact fp32 products)
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt32 -o p -- python3 bench.py --cpu-seconds 0 --no-configs --no-extras --precision fp32 --steps 3 --warmup 1 > $O/bench_fp32_noextras_under_rocprof.json 2>> $O/bench.err
cp $(find $O/kt32 -name "*kernel_stats.csv" | head -1) $O/bench_fp32_noextras_kernel_stats.csv; rm -rf $O/kt32
CMD="python3 bench.py --clips 64 --steps 1 --warmup 0 --cpu-seconds 0 --no-configs --no-extras"
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_f -o p --output-format csv -- $CMD > $O/pmc_f.log 2>&1
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_w -o p --output-format csv -- $CMD > $O/pmc_w.log 2>&1
python tools/summarize_pmc.py $O/pmc_f $O/pmc_w $O/pmc_traffic_bf16x3.json 64 $FAM "the MFMA convolution launches of ONE {clips}-clip UNet eval forward (bf16x3)" '(, 1(, (false|true)(, [0-9]+)?(, (false|true))?)?>$)|(conv_wd16_kernel)|(conv_ws64_kernel)|(conv_up_kernel)' > $O/pmc_traffic.log 2>&1
rm -rf $O/pmc_f $O/pmc_w
timeout -k 10 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d $O/pmc_s1 -o p --output-format csv -- $CMD > $O/pmc_s1.log 2>&1
timeout -k 10 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU -d $O/pmc_s2 -o p --output-format csv -- $CMD > $O/pmc_s2.log 2>&1
timeout -k 10 400 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum SQ_INSTS_SMEM -d $O/pmc_s3 -o p --output-format csv -- $CMD > $O/pmc_s3.log 2>&1
python tools/summarize_sq.py $O/pmc_s1 $FAM $O/pmc_sq_pass1.json > $O/pmc_sq1.txt 2>&1
python tools/summarize_sq.py $O/pmc_s2 $FAM $O/pmc_sq_pass2.json > $O/pmc_sq2.txt 2>&1
python tools/summarize_sq.py $O/pmc_s3 $FAM $O/pmc_sq_pass3.json > $O/pmc_sq3.txt 2>&1
python tools/sq_table.py $O/pmc_sq_pass1.json $O/pmc_sq_pass2.json > $O/pmc_sq_table.md 2>&1
rm -rf $O/pmc_s1 $O/pmc_s2 $O/pmc_s3
python tools/exp_upconv.py --clips 64 2>&1 | grep -v amdgpu > $O/upconv_levels.txt
python tools/exp_upconv.py --clips 64 --precision 0 --reps 3 2>&1 | grep -v amdgpu > $O/upconv_levels_fp32.txt
timeout -k 10 300 python bench.py --mode train --precision bf16 --wgrad bf16 --augment --steps 20 --warmup 5 > $O/train_step_bf16_bench_line.json 2>> $O/bench.err
timeout -k 10 300 python bench.py --mode train --precision bf16 --wgrad bf16 --steps 20 --warmup 5 > $O/train_step_bf16_premixed_bench_line.json 2>> $O/bench.err
timeout -k 10 300 python bench.py --mode demucs --steps 20 --warmup 5 > $O/demucs_bench_line.json 2>> $O/bench.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktt -o p -- python3 bench.py --mode train --precision bf16 --wgrad bf16 --augment --steps 10 --warmup 3 --cpu-seconds 0 > $O/train_step_bf16_bench_under_rocprof.json 2>> $O/bench.err
cp $(find $O/ktt -name "*kernel_stats.csv" | head -1) $O/train_step_bf16_kernel_stats.csv; rm -rf $O/ktt
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktd -o p -- python3 bench.py --mode demucs --steps 10 --warmup 3 --cpu-seconds 0 > $O/demucs_bench_under_rocprof.json 2>> $O/bench.err
cp $(find $O/ktd -name "*kernel_stats.csv" | head -1) $O/demucs_kernel_stats.csv; rm -rf $O/ktd
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt2 -o p -- python3 bench.py --no-unet --steps 20 --warmup 3 > $O/config2_bench_under_rocprof.json 2>> $O/bench.err
cp $(find $O/kt2 -name "*kernel_stats.csv" | head -1) $O/config2_kernel_stats.csv; rm -rf $O/kt2
python tools/exp_lstm.py 2>&1 | grep -v amdgpu > $O/demucs_lstm_step.txt
python tools/time_small_kernels.py 2>&1 | grep -v amdgpu > $O/small_kernels.txt
ls $O; tail -2 $O/pmc_traffic.log; cat $O/pmc_sq_table.md
