L=musicfpaugment_amd/libmfpa_exp.so
for i in 1 2; do
for r in 0 1; do
MFPA_CONV_PLAIN_ROWS=$r python bench.py --mode train --precision bf16 --steps 10 --warmup 3 --lib $L 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('plain_rows=$r', d['value'], d['ms_per_step'], d['config']['loss_last'], d['roofline']['kernel_ms_per_step'])"
done
done
