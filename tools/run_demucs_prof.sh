#!/bin/bash
export TMPDIR=/tmp; O=gpurun_out/demucs_prof; mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o dm -- python3 bench.py --mode demucs --no-configs --steps 3 --warmup 1 > $O/bench.log 2>&1
cp $(find $O/kt -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv; rm -rf $O/kt
head -16 $O/kernel_stats.csv | cut -c1-150
