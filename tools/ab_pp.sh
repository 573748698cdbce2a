for i in 1 2; do
python bench.py --mode train --precision bf16 --steps 10 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('base ', d['value'], d['ms_per_step'], d['config']['loss_last'], d['roofline']['kernel_ms_per_step'])"
python bench.py --mode train --precision bf16 --steps 10 --warmup 3 --lib musicfpaugment_amd/libmfpa_pp.so 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('pp   ', d['value'], d['ms_per_step'], d['config']['loss_last'], d['roofline']['kernel_ms_per_step'])"
done
