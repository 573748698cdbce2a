#!/usr/bin/env python3
"""Top rows of a rocprofv3 --kernel-trace --stats CSV found under a directory: usage kstats.py <dir> [steps] [rows]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
n = int(sys.argv[3]) if len(sys.argv) > 3 else 14
for r in list(csv.DictReader(open(f)))[:n]:
    print("%8.2f ms/step %5d calls %9.1f us  %s" % (float(r["TotalDurationNs"]) / 1e6 / steps, int(r["Calls"]), float(r["AverageNs"]) / 1e3, r["Name"][:110]))
