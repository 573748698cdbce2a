#!/bin/bash
# same-box A/B of the Demucs forward: persistent LSTM variants vs the per-step kernels
export TMPDIR=/tmp; mkdir -p gpurun_out/demucs
timeout -k 10 300 python -m pytest tests/test_gpu_demucs.py -x -q > gpurun_out/demucs/tests.log 2>&1 || { tail -30 gpurun_out/demucs/tests.log; exit 1; }
tail -3 gpurun_out/demucs/tests.log
EXP=musicfpaugment_amd/libmfpa_exp.so
run() { timeout -k 10 200 python bench.py --mode demucs --no-configs --lib $EXP 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'])" || exit 1; }
for i in 1 2; do
MFPA_LSTM_COH=0 run "seq-inv  "
MFPA_LSTM_COH=1 run "seq-sc1  "
MFPA_LSTM_SEQ=0 run "steps    "
done
