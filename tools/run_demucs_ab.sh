#!/bin/bash
# same-box A/B of the Demucs forward: pipelined 256 x 128 GEMM vs the 128 x 128 kernel
export TMPDIR=/tmp; mkdir -p gpurun_out/demucs
timeout -k 10 300 python -m pytest tests/test_gpu_demucs.py -x -q > gpurun_out/demucs/tests.log 2>&1 || { tail -30 gpurun_out/demucs/tests.log; exit 1; }
tail -3 gpurun_out/demucs/tests.log
EXP=musicfpaugment_amd/libmfpa_exp.so
MFPA_GEMM_PIPE=1 timeout -k 10 200 python tools/exp_demucs_layers.py --lib $EXP > gpurun_out/demucs/layers_pipe.txt 2>&1 || exit 1
MFPA_GEMM_PIPE=0 timeout -k 10 200 python tools/exp_demucs_layers.py --lib $EXP > gpurun_out/demucs/layers_wide.txt 2>&1 || exit 1
paste <(cut -c1-48 gpurun_out/demucs/layers_pipe.txt) <(cut -c34-48 gpurun_out/demucs/layers_wide.txt) | tail -24
run() { timeout -k 10 200 python bench.py --mode demucs --no-configs --lib $EXP 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'])" || exit 1; }
for i in 1 2; do
MFPA_GEMM_PIPE=1 run "pipe "
MFPA_GEMM_PIPE=0 run "wide "
done
