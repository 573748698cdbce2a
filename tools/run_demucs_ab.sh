#!/bin/bash
# same-box A/B of the Demucs forward: fused first encoder level on / off
export TMPDIR=/tmp; mkdir -p gpurun_out/demucs
timeout -k 10 300 python -m pytest tests/test_gpu_demucs.py -x -q > gpurun_out/demucs/tests.log 2>&1 || { tail -30 gpurun_out/demucs/tests.log; exit 1; }
tail -3 gpurun_out/demucs/tests.log
for i in 1 2; do
timeout -k 10 200 python bench.py --mode demucs --no-configs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fused', d['value'], d['ms_per_step'])" || exit 1
timeout -k 10 200 python - <<'PY' || exit 1
import subprocess, sys, json, os
code = "import sys; sys.argv=['bench.py','--mode','demucs','--no-configs']; from musicfpaugment_amd import ops_demucs as D; D.FUSE_FIRST_LEVEL=False; import runpy; runpy.run_path('bench.py', run_name='__main__')"
out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True).stdout.strip().splitlines()[-1]
d = json.loads(out); print('two  ', d['value'], d['ms_per_step'])
PY
done
