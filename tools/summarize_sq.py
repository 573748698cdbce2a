#!/usr/bin/env python3
"""Per-kernel means of the SQ / GRBM counters of one `rocprofv3 --pmc ...` pass.

usage: summarize_sq.py <dir> [family,family,...] [out.json]
Rows of one dispatch (one per XCD / shader engine) are summed; then the mean per dispatch of every counter is printed per
kernel (template arguments kept, namespaces dropped), with the dispatch's wall time from its timestamps.  Units as
MI355X_MICROARCH.md states: SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves;
SQ_VALU_MFMA_BUSY_CYCLES and SQ_BUSY_CYCLES count cycles summed over SIMDs / SEs; GRBM_GUI_ACTIVE is summed over the 8 XCDs."""
import collections, csv, glob, json, sys

d = sys.argv[1]
fams = tuple(sys.argv[2].split(",")) if len(sys.argv) > 2 else ("conv_mfma_kernel", "convT_mfma_kernel")
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
disp = collections.defaultdict(lambda: collections.defaultdict(float))
meta = {}
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if not any(x in n for x in fams):
        continue
    k = r["Dispatch_Id"]
    disp[k][r["Counter_Name"]] += float(r["Counter_Value"])
    meta[k] = (n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0],
               (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["VGPR_Count"], r["LDS_Block_Size"], r["Scratch_Size"])
per = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for k, c in disp.items():
    name = meta[k][0]
    cnt[name] += 1
    per[name]["us"] += meta[k][1]
    for cn, v in c.items():
        per[name][cn] += v
out = {}
for name in sorted(per, key=lambda n: -per[n]["us"]):
    n = cnt[name]
    row = {cn: v / n for cn, v in per[name].items()}
    row["launches"] = n
    k0 = next(k for k in meta if meta[k][0] == name)
    row["vgpr"], row["lds"], row["scratch"] = meta[k0][2], meta[k0][3], meta[k0][4]
    out[name] = row
    print(name, json.dumps({k: (round(v, 1) if isinstance(v, float) else v) for k, v in row.items()}))
if len(sys.argv) > 3:
    json.dump(out, open(sys.argv[3], "w"), indent=1)
