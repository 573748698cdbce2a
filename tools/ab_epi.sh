for i in 1 2; do
python tools/exp_c64.py --lib musicfpaugment_amd/libmfpa_epi0.so | grep -v "amdgpu\|DBG" | sed 's/^/epi0 /'
python tools/exp_c64.py | grep -v "amdgpu\|DBG" | sed 's/^/epi1 /'
done
for i in 1 2; do
python bench.py --steps 8 --warmup 2 --no-configs --cpu-seconds 0 --lib musicfpaugment_amd/libmfpa_epi0.so 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('epi0', d['value'], d['roofline']['kernel_ms_per_step'])"
python bench.py --steps 8 --warmup 2 --no-configs --cpu-seconds 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('epi1', d['value'], d['roofline']['kernel_ms_per_step'])"
done
