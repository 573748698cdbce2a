# Round-5 profile set of the FINAL binary: the driver's bench line, rocprofv3 kernel stats of the same command, PMC traffic + SQ counters of the
# conv family, train (plain bf16 and bf16x3) / demucs lines, config 2 stats.  Outputs under gpurun_out/r05p; copy what is judged into profiles/r05_*.
export TMPDIR=/tmp
O=gpurun_out/r05p; mkdir -p $O
timeout -k 10 700 python bench.py --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench.err
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o p -- python3 bench.py --cpu-seconds 0 --no-configs > $O/bench_under_rocprof.json 2>> $O/bench.err
cp $(find $O/kt -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv; rm -rf $O/kt
# the same trace WITHOUT the line's extras (other-precision leg, 8-clip parity passes, under-load burst): every traced launch is a 128-clip pass of the timed chain
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt0 -o p -- python3 bench.py --cpu-seconds 0 --no-configs --no-extras --steps 5 --warmup 1 > $O/bench_noextras_under_rocprof.json 2>> $O/bench.err
cp $(find $O/kt0 -name "*kernel_stats.csv" | head -1) $O/bench_noextras_kernel_stats.csv; rm -rf $O/kt0
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_f -o p --output-format csv -- python3 bench.py --clips 64 --steps 1 --warmup 0 --cpu-seconds 0 --no-configs --no-extras > $O/pmc_f.log 2>&1
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_w -o p --output-format csv -- python3 bench.py --clips 64 --steps 1 --warmup 0 --cpu-seconds 0 --no-configs --no-extras > $O/pmc_w.log 2>&1
python tools/summarize_pmc.py $O/pmc_f $O/pmc_w $O/pmc_traffic_bf16x3.json 64 conv_mfma_kernel,convT_mfma_kernel,conv_wd16_kernel,conv_ws64_kernel "the MFMA convolution launches of ONE {clips}-clip UNet eval forward (bf16x3)" '(, 1(, (false|true)(, [0-9]+)?(, (false|true))?)?>$)|(conv_wd16_kernel)|(conv_ws64_kernel)' > $O/pmc_traffic.log 2>&1
rm -rf $O/pmc_f $O/pmc_w
timeout -k 10 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d $O/pmc_s1 -o p --output-format csv -- python3 bench.py --clips 64 --steps 1 --warmup 0 --cpu-seconds 0 --no-configs --no-extras > $O/pmc_s1.log 2>&1
timeout -k 10 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU -d $O/pmc_s2 -o p --output-format csv -- python3 bench.py --clips 64 --steps 1 --warmup 0 --cpu-seconds 0 --no-configs --no-extras > $O/pmc_s2.log 2>&1
timeout -k 10 400 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum SQ_INSTS_SMEM -d $O/pmc_s3 -o p --output-format csv -- python3 bench.py --clips 64 --steps 1 --warmup 0 --cpu-seconds 0 --no-configs --no-extras > $O/pmc_s3.log 2>&1
python tools/summarize_sq.py $O/pmc_s1 conv_mfma_kernel,convT_mfma_kernel,conv_wd16_kernel,conv_ws64_kernel $O/pmc_sq_pass1.json > $O/pmc_sq1.txt 2>&1
python tools/summarize_sq.py $O/pmc_s2 conv_mfma_kernel,convT_mfma_kernel,conv_wd16_kernel,conv_ws64_kernel $O/pmc_sq_pass2.json > $O/pmc_sq2.txt 2>&1
python tools/summarize_sq.py $O/pmc_s3 conv_mfma_kernel,convT_mfma_kernel,conv_wd16_kernel,conv_ws64_kernel $O/pmc_sq_pass3.json > $O/pmc_sq3.txt 2>&1
python tools/sq_table.py $O/pmc_sq_pass1.json $O/pmc_sq_pass2.json > $O/pmc_sq_table.md 2>&1
rm -rf $O/pmc_s1 $O/pmc_s2 $O/pmc_s3
timeout -k 10 300 python bench.py --mode train --precision bf16 --wgrad bf16 --augment --steps 20 --warmup 5 > $O/train_step_bf16_bench_line.json 2>> $O/bench.err
timeout -k 10 300 python bench.py --mode train --precision bf16x3 --wgrad bf16 --augment --steps 20 --warmup 5 > $O/train_step_bf16x3_bench_line.json
timeout -k 10 300 python bench.py --mode train --precision bf16 --wgrad bf16 --augment --no-z16 --steps 20 --warmup 5 > $O/train_step_bf16_f32act_bench_line.json 2>> $O/bench.err
timeout -k 10 300 python bench.py --mode train --precision bf16 --wgrad bf16 --steps 20 --warmup 5 > $O/train_step_bf16_premixed_bench_line.json 2>> $O/bench.err 2>> $O/bench.err
timeout -k 10 300 python bench.py --mode demucs --steps 20 --warmup 5 > $O/demucs_bench_line.json 2>> $O/bench.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktt -o p -- python3 bench.py --mode train --precision bf16 --wgrad bf16 --augment --steps 10 --warmup 3 --cpu-seconds 0 > $O/train_step_bf16_bench_under_rocprof.json 2>> $O/bench.err
cp $(find $O/ktt -name "*kernel_stats.csv" | head -1) $O/train_step_bf16_kernel_stats.csv; rm -rf $O/ktt
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt2 -o p -- python3 bench.py --no-unet --steps 20 --warmup 3 > $O/config2_bench_under_rocprof.json 2>> $O/bench.err
cp $(find $O/kt2 -name "*kernel_stats.csv" | head -1) $O/config2_kernel_stats.csv; rm -rf $O/kt2
python tools/exp_conv.py --both --reps 5 > $O/conv_layers_lds_vs_direct.txt 2>&1
python tools/exp_c64.py > $O/c64_layers.txt 2>&1
python tools/time_small_kernels.py > $O/small_kernels.txt 2>&1
python tools/ab_pick.py > $O/pick_stage.txt 2>&1
python tools/ab_dejavu.py > $O/dejavu_stages.txt 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt3 -o p -- python3 bench.py --no-unet --picker dejavu --steps 20 --warmup 3 --cpu-seconds 0 --no-configs > $O/config2_dejavu_bench_under_rocprof.json 2>> $O/bench.err
cp $(find $O/kt3 -name "*kernel_stats.csv" | head -1) $O/config2_dejavu_kernel_stats.csv; rm -rf $O/kt3
bash tools/run_config2_sq.sh > $O/config2_sq.log 2>&1
cp gpurun_out/config2_sq/sq1_256.json $O/config2_sq_pass1.json; cp gpurun_out/config2_sq/sq2_256.json $O/config2_sq_pass2.json
ls $O; tail -2 $O/pmc_traffic.log
python tools/exp_ws_timeline.py musicfpaugment_amd/libmfpa_ws_0_MFPA_WS_STAMPS1.so > $O/ws_timeline.txt 2>&1
for s_ in 1 3; do python bench.py --mode infer --no-unet --clips 256 --steps 60 --warmup 6 --batch-streams $s_ --no-extras --cpu-seconds 0 --no-configs 2>/dev/null; done > $O/config2_batch_streams_lines.txt 2>&1
cd /tmp; rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/ktr -o tr -- python3 $GRAFT_REPO_ROOT/bench.py --mode train --precision bf16 --wgrad bf16 --steps 4 --warmup 2 --no-configs --no-extras --cpu-seconds 0 > /dev/null 2>&1; cd $GRAFT_REPO_ROOT; python tools/train_trace.py $O/ktr/tr_kernel_trace.csv > $O/train_step_launches.txt 2>&1; rm -rf $O/ktr
for v in "" _old; do echo "== libmfpa$v.so"; python tools/exp_c64.py --lib musicfpaugment_amd/libmfpa$v.so 2>&1 | grep -v amdgpu; done > $O/c64_new_vs_old.txt 2>&1
for f in "" "--lib musicfpaugment_amd/libmfpa_old.so" "" "--lib musicfpaugment_amd/libmfpa_old.so"; do echo "== bench.py $f"; python bench.py --steps 10 --warmup 3 --no-configs --cpu-seconds 0 $f 2>/dev/null; done > $O/bench_new_vs_old.txt 2>&1
