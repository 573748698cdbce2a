#!/bin/bash
export TMPDIR=/tmp; O=gpurun_out/demucs_prof; mkdir -p $O
EXP=musicfpaugment_amd/libmfpa_exp.so
MFPA_LSTM_COH=0 timeout -k 10 120 python tools/exp_lstm.py --lib $EXP || exit 1
MFPA_LSTM_COH=1 timeout -k 10 120 python tools/exp_lstm.py --lib $EXP || exit 1
timeout -k 10 120 python tools/exp_lstm.py --lib $EXP --persistent 0 || exit 1
MFPA_LSTM_COH=1 timeout -k 10 120 python tools/exp_lstm.py --lib $EXP --clips 64 || exit 1
timeout -k 10 120 python tools/exp_lstm.py --lib $EXP --clips 64 --persistent 0 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o dm -- python3 bench.py --mode demucs --no-configs --steps 3 --warmup 1 > $O/bench.log 2>&1
cp $(find $O/kt -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv; rm -rf $O/kt
head -22 $O/kernel_stats.csv | cut -c1-180
