for i in 1 2; do
python tools/exp_c64.py 2>/dev/null | grep -v "amdgpu\|DBG\|inc.3\|convT" | sed 's/^/base   /'
python tools/exp_c64.py --lib musicfpaugment_amd/libmfpa_spread.so 2>/dev/null | grep -v "amdgpu\|DBG\|inc.3\|convT" | sed 's/^/spread /'
done
for i in 1 2; do
python bench.py --steps 8 --warmup 2 --no-configs --cpu-seconds 0 --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('base  ', d['value'], d['roofline']['kernel_ms_per_step'])"
python bench.py --steps 8 --warmup 2 --no-configs --cpu-seconds 0 --no-extras --lib musicfpaugment_amd/libmfpa_spread.so 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('spread', d['value'], d['roofline']['kernel_ms_per_step'])"
done
