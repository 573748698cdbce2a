# SQ counters of ONE plain-bf16 train step's MFMA kernels (forward / input-gradient convolutions, weight gradients): two --pmc passes of
# `bench.py --mode train --precision bf16 --wgrad bf16 --steps 1 --warmup 1` (pre-mixed), summarised by tools/summarize_sq.py.
export TMPDIR=/tmp
O=gpurun_out/train_sq; mkdir -p $O
CMD="python3 bench.py --mode train --precision bf16 --wgrad bf16 --steps 1 --warmup 1 --cpu-seconds 0 --no-configs --no-extras"
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d $O/s1 -o p --output-format csv -- $CMD > $O/s1.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU -d $O/s2 -o p --output-format csv -- $CMD > $O/s2.log 2>&1
FAM=conv_wd16_kernel,wgrad_bf16_kernel,wgrad_bf16x3_kernel,convT_mfma_kernel,conv_mfma_kernel
python tools/summarize_sq.py $O/s1 $FAM $O/pass1.json > $O/sq1.txt 2>&1
python tools/summarize_sq.py $O/s2 $FAM $O/pass2.json > $O/sq2.txt 2>&1
rm -rf $O/s1 $O/s2
python - <<'PY'
import json
a, b = json.load(open("gpurun_out/train_sq/pass1.json")), json.load(open("gpurun_out/train_sq/pass2.json"))
print("| kernel | launches | us (sum) | MFMA busy | clock GHz | parked | issue-stalled | issuing | LDS issue | VALU / MFMA | LDS / MFMA | LDS conflict / active | VGPRs |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|---|")
for k, r in sorted(a.items(), key=lambda kv: -kv[1]["us"] * kv[1]["launches"]):
    r2 = b.get(k, {})
    wc, gui = r["SQ_WAVE_CYCLES"], r["GRBM_GUI_ACTIVE"] / 8
    mf = max(r2.get("SQ_INSTS_MFMA", 1), 1)
    print(f"| `{k[:70]}` | {r['launches']} | {r['us'] * r['launches']:.0f} | {r['SQ_VALU_MFMA_BUSY_CYCLES'] / (gui * 1024):.3f} | {gui / (r['us'] * 1e-6) / 1e9:.2f} | "
          f"{r['SQ_WAIT_ANY'] / wc:.3f} | {r['SQ_WAIT_INST_ANY'] / wc:.3f} | {r['SQ_ACTIVE_INST_ANY'] / wc:.3f} | {r['SQ_WAIT_INST_LDS'] / wc:.3f} | "
          f"{r2.get('SQ_INSTS_VALU', 0) / mf:.2f} | {r2.get('SQ_INSTS_LDS', 0) / mf:.2f} | {r2.get('SQ_LDS_BANK_CONFLICT', 0) / max(r2.get('SQ_LDS_IDX_ACTIVE', 1), 1):.3f} | {r['vgpr']} |")
PY
