#!/usr/bin/env python3
"""One train step's launches in order, from a rocprofv3 --kernel-trace csv of `bench.py --mode train --steps S --warmup W` (no configs):
the LAST step's launches (split at stft_kernel pairs), name shortened, duration, gap to the previous launch's end.
usage: train_trace.py <kernel_trace.csv> [nsteps_in_trace]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    m = re.match(r"([A-Za-z0-9_:]+(<[^>]*>)?)", n)
    return m.group(1) if m else n[:60]
# a step starts at the first of its two stft_kernel launches (clean + augmented): take the last complete step
idx = [i for i, r in enumerate(rows) if "stft_kernel" in r["Kernel_Name"]]
starts = idx[::2] if len(idx) >= 2 else [0]
a = starts[-2] if len(starts) >= 2 else starts[-1]
b = starts[-1] if len(starts) >= 2 else len(rows)
prev_end = None
tot = 0
for r in rows[a:b]:
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (st - prev_end) / 1e3 if prev_end else 0.0
    tot += (en - st)
    print(f"{short(r['Kernel_Name']):70s} {(en - st) / 1e3:9.1f} us  gap {gap:7.1f}  grid {r.get('Grid_Size_X', r.get('Grid_Size', '?'))}")
    prev_end = en
print(f"launches {b - a}, kernel time {tot / 1e6:.3f} ms, span {(int(rows[b - 1]['End_Timestamp']) - int(rows[a]['Start_Timestamp'])) / 1e6:.3f} ms")
