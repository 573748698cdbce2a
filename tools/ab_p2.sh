for i in 1 2; do
python tools/exp_conv.py --both --reps 5 2>/dev/null | grep -v amdgpu | awk '{print "base ", $1, $2, $3, $9, $10}' | head -14
python tools/exp_conv.py --both --reps 5 --lib musicfpaugment_amd/libmfpa_pr.so 2>/dev/null | grep -v amdgpu | awk '{print "prows", $1, $2, $3, $9, $10}' | head -14
done
for i in 1 2; do
python bench.py --steps 8 --warmup 2 --no-configs --cpu-seconds 0 --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('base  ', d['value'], d['roofline']['kernel_ms_per_step'])"
python bench.py --steps 8 --warmup 2 --no-configs --cpu-seconds 0 --no-extras --lib musicfpaugment_amd/libmfpa_pr.so 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('prows ', d['value'], d['roofline']['kernel_ms_per_step'])"
done
