#!/usr/bin/env python3
"""Sum FETCH_SIZE / WRITE_SIZE over the conv_mfma_kernel launches of two rocprofv3 --pmc passes (one counter per pass).

usage: summarize_pmc.py <fetch_dir> <write_dir> <out.json> [clips] [family,family,...] [label] [regex]
(default families: the UNet convolution kernels; e.g. "gemm_,lstm_step" for the Demucs forward; `regex`: only kernels whose
short name matches, e.g. ", 1(, (false|true))?>$" = the PREC 1 (bf16x3) instantiations)
Corrections as MI355X_MICROARCH.md prescribes for gfx950: FETCH_SIZE is reported in KiB-like units of 64 B per 128-B
request for 16-B/lane coalesced reads -> raw bytes x 2; WRITE_SIZE is used as reported."""
import csv, glob, json, re, sys, collections

def load(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    per = collections.defaultdict(lambda: [0, 0.0])
    # several rows per dispatch (one per XCD / dimension): sum by dispatch id first
    disp = collections.defaultdict(float); name = {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        disp[r["Dispatch_Id"]] += float(r["Counter_Value"]); name[r["Dispatch_Id"]] = r["Kernel_Name"]
    for k, v in disp.items():
        for fam in FAMILIES:
            if fam in name[k]:
                short = name[k].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
                if REGEX is not None and not REGEX.search(short):
                    break
                per[short][0] += 1; per[short][1] += v
                break
    return per

fd, wd, out = sys.argv[1:4]
clips = int(sys.argv[4]) if len(sys.argv) > 4 else 64
FAMILIES = tuple(sys.argv[5].split(",")) if len(sys.argv) > 5 else ("conv_mfma_kernel", "convT_mfma_kernel")
REGEX = re.compile(sys.argv[7]) if len(sys.argv) > 7 else None
LABEL = sys.argv[6] if len(sys.argv) > 6 else "the MFMA convolution launches of ONE {clips}-clip UNet eval forward (bf16x3)"
F, W = load(fd, "FETCH_SIZE"), load(wd, "WRITE_SIZE")
unit = 1024.0      # rocprofv3 reports both counters in KiB
fetch_raw = sum(v[1] for v in F.values()) * unit
write = sum(v[1] for v in W.values()) * unit
res = {"what": "HBM-side traffic of " + LABEL.format(clips=clips) + ", rocprofv3 --pmc, one counter per pass", "launches": sum(v[0] for v in F.values()),
       "FETCH_SIZE_bytes_raw": fetch_raw, "FETCH_SIZE_bytes_corrected_x2": 2 * fetch_raw, "WRITE_SIZE_bytes": write,
       "per_256_clip_step_bytes": (2 * fetch_raw + write) * 256 / clips,
       "per_kernel": {k: {"launches": F[k][0], "fetch_raw": F[k][1] * unit, "write": W[k][1] * unit} for k in F}}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps({k: res[k] for k in ("launches", "FETCH_SIZE_bytes_corrected_x2", "WRITE_SIZE_bytes", "per_256_clip_step_bytes")}))
