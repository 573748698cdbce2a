#!/usr/bin/env python3
"""Markdown table of the SQ counter passes of the conv family (tools/summarize_sq.py JSONs): usage sq_table.py pass1.json pass2.json"""
import json, re, sys
PREC1 = re.compile(r", 1(, (false|true)(, [0-9]+)?(, (false|true))?)?>$")      # the PREC template argument
a, b = json.load(open(sys.argv[1])), json.load(open(sys.argv[2]))
print("| kernel | launches | µs | MFMA busy | clock GHz | parked | issue-stalled | issuing | of which LDS issue | VALU / MFMA | SALU / MFMA | LDS / MFMA | LDS bank conflict ÷ LDS active | VGPRs |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|---|---|")
for k in a:
    r, r2 = a[k], b.get(k, {})
    if not PREC1.search(k) and "conv_wd16_kernel" not in k and "conv_ws64_kernel" not in k and "conv_up_kernel" not in k:      # PREC 1 instantiations (and the bf16x3-only 16 x 16 x 32 kernel)
        continue
    wc, gui = r["SQ_WAVE_CYCLES"], r["GRBM_GUI_ACTIVE"] / 8
    mf = max(r2.get("SQ_INSTS_MFMA", 1), 1)
    print(f"| `{k}` | {r['launches']} | {r['us']:.0f} | {r['SQ_VALU_MFMA_BUSY_CYCLES'] / (gui * 1024):.3f} | {gui / (r['us'] * 1e-6) / 1e9:.2f} | "
          f"{r['SQ_WAIT_ANY'] / wc:.3f} | {r['SQ_WAIT_INST_ANY'] / wc:.3f} | {r['SQ_ACTIVE_INST_ANY'] / wc:.3f} | {r['SQ_WAIT_INST_LDS'] / wc:.3f} | "
          f"{r2.get('SQ_INSTS_VALU', 0) / mf:.2f} | {r2.get('SQ_INSTS_SALU', 0) / mf:.2f} | {r2.get('SQ_INSTS_LDS', 0) / mf:.2f} | "
          f"{r2.get('SQ_LDS_BANK_CONFLICT', 0) / max(r2.get('SQ_LDS_IDX_ACTIVE', 1), 1):.3f} | {r['vgpr']} |")
