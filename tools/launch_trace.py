#!/usr/bin/env python3
"""All launches of the LAST `n` repetitions in a rocprofv3 --kernel-trace csv, in order: name (shortened), duration, gap to the previous launch's end.
usage: launch_trace.py <kernel_trace.csv> <first-kernel-substring of a repetition> [n=1]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
key = sys.argv[2]
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    m = re.match(r"([A-Za-z0-9_:]+(<[^>]*>)?)", n)
    return m.group(1) if m else n[:60]
idx = [i for i, r in enumerate(rows) if key in r["Kernel_Name"] and (i == 0 or key not in rows[i - 1]["Kernel_Name"])]
a, b = (idx[-2], idx[-1]) if len(idx) >= 2 else (0, len(rows))
prev_end, tot = None, 0
for r in rows[a:b]:
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (st - prev_end) / 1e3 if prev_end else 0.0
    tot += en - st
    print(f"{short(r['Kernel_Name']):72s} {(en - st) / 1e3:9.1f} us  gap {gap:7.1f}")
    prev_end = max(prev_end or 0, en)
print(f"launches {b - a}, kernel time {tot / 1e6:.3f} ms, span {(max(int(r['End_Timestamp']) for r in rows[a:b]) - int(rows[a]['Start_Timestamp'])) / 1e6:.3f} ms")
