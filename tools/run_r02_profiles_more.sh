#!/bin/bash
# Round-2 profile set, part 2: the Demucs forward (config 5) and the two training steps (config 4 and the Demucs step).
export TMPDIR=/tmp
O=gpurun_out/r02p2; mkdir -p $O
timeout -k 10 300 python bench.py --mode demucs --no-configs > $O/demucs_bench_line.json 2> $O/err.log
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o p -- python3 bench.py --mode demucs --no-configs > $O/demucs_bench_under_rocprof.json 2>> $O/err.log
cp $(find $O/kt -name "*kernel_stats.csv" | head -1) $O/demucs_kernel_stats.csv; rm -rf $O/kt
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_f -o p --output-format csv -- python3 bench.py --mode demucs --no-configs --steps 1 --warmup 0 > $O/pmc_f.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_w -o p --output-format csv -- python3 bench.py --mode demucs --no-configs --steps 1 --warmup 0 > $O/pmc_w.log 2>&1
python tools/summarize_pmc.py $O/pmc_f $O/pmc_w $O/pmc_traffic_demucs.json 256 gemm_,lstm_,c1_glu_kernel,glu_convT_c1_kernel "the GEMM-family and LSTM launches of ONE {clips}-clip Demucs forward (bf16x3)" > $O/pmc_traffic.log 2>&1
rm -rf $O/pmc_f $O/pmc_w
timeout -k 10 300 python bench.py --mode train --no-configs > $O/train_step_bench_line.json 2>> $O/err.log
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o p -- python3 bench.py --mode train --no-configs > $O/train_step_bench_under_rocprof.json 2>> $O/err.log
cp $(find $O/kt -name "*kernel_stats.csv" | head -1) $O/train_step_kernel_stats.csv; rm -rf $O/kt
timeout -k 10 300 python bench.py --mode demucs-train --no-configs > $O/demucs_train_bench_line.json 2>> $O/err.log
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o p -- python3 bench.py --mode demucs-train --no-configs > $O/demucs_train_bench_under_rocprof.json 2>> $O/err.log
cp $(find $O/kt -name "*kernel_stats.csv" | head -1) $O/demucs_train_kernel_stats.csv; rm -rf $O/kt
ls $O; tail -2 $O/pmc_traffic.log; for f in demucs_bench_line train_step_bench_line demucs_train_bench_line; do tail -1 $O/$f.json | cut -c1-160; done
