# Round-3 profile set of the FINAL binary: default bench line, rocprofv3 kernel stats of the same command, PMC traffic + SQ counters of the
# conv family, train / demucs lines, small-kernel timings.  Outputs under gpurun_out/r03p; copy what is judged into profiles/r03_*.
export TMPDIR=/tmp
O=gpurun_out/r03p; mkdir -p $O
timeout -k 10 500 python bench.py --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench.err
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o p -- python3 bench.py --cpu-seconds 0 --no-configs > $O/bench_under_rocprof.json 2>> $O/bench.err
cp $O/kt/p_kernel_stats.csv $O/bench_kernel_stats.csv 2>/dev/null || cp $(find $O/kt -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv
rm -rf $O/kt
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_f -o p --output-format csv -- python3 bench.py --clips 64 --steps 1 --warmup 0 --cpu-seconds 0 --no-configs > $O/pmc_f.log 2>&1
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_w -o p --output-format csv -- python3 bench.py --clips 64 --steps 1 --warmup 0 --cpu-seconds 0 --no-configs > $O/pmc_w.log 2>&1
python tools/summarize_pmc.py $O/pmc_f $O/pmc_w $O/pmc_traffic_bf16x3.json 64 conv_mfma_kernel,convT_mfma_kernel,conv_wd16_kernel "the MFMA convolution launches of ONE {clips}-clip UNet eval forward (bf16x3)" '(, 1(, (false|true)(, [0-9]+)?(, (false|true))?)?>$)|(conv_wd16_kernel)' > $O/pmc_traffic.log 2>&1
rm -rf $O/pmc_f $O/pmc_w
timeout -k 10 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d $O/pmc_s1 -o p --output-format csv -- python3 bench.py --clips 64 --steps 1 --warmup 0 --cpu-seconds 0 --no-configs > $O/pmc_s1.log 2>&1
timeout -k 10 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU -d $O/pmc_s2 -o p --output-format csv -- python3 bench.py --clips 64 --steps 1 --warmup 0 --cpu-seconds 0 --no-configs > $O/pmc_s2.log 2>&1
python tools/summarize_sq.py $O/pmc_s1 conv_mfma_kernel,convT_mfma_kernel,conv_wd16_kernel $O/pmc_sq_pass1.json > $O/pmc_sq1.txt 2>&1
python tools/summarize_sq.py $O/pmc_s2 conv_mfma_kernel,convT_mfma_kernel,conv_wd16_kernel $O/pmc_sq_pass2.json > $O/pmc_sq2.txt 2>&1
python tools/sq_table.py $O/pmc_sq_pass1.json $O/pmc_sq_pass2.json > $O/pmc_sq_table.md 2>&1
rm -rf $O/pmc_s1 $O/pmc_s2
timeout -k 10 300 python bench.py --mode train --steps 20 --warmup 5 > $O/train_step_bench_line.json 2>> $O/bench.err
timeout -k 10 300 python bench.py --mode demucs --steps 20 --warmup 5 > $O/demucs_bench_line.json 2>> $O/bench.err
timeout -k 10 300 python bench.py --mode demucs-train --steps 20 --warmup 5 > $O/demucs_train_bench_line.json 2>> $O/bench.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktt -o p -- python3 bench.py --mode train --steps 10 --warmup 3 --cpu-seconds 0 > $O/train_step_bench_under_rocprof.json 2>> $O/bench.err
cp $(find $O/ktt -name "*kernel_stats.csv" | head -1) $O/train_step_kernel_stats.csv; rm -rf $O/ktt
python tools/exp_wgrad.py > $O/wgrad_layers.txt 2>&1
python tools/time_small_kernels.py 256 > $O/small_kernels.txt 2>&1
python tools/exp_conv.py --both --reps 5 > $O/conv_layers_lds_vs_direct.txt 2>&1
ls $O; cat $O/pmc_traffic.log | tail -2
