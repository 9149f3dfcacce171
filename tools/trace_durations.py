"""Per-kernel duration percentiles from a rocprofv3 --kernel-trace csv directory.  usage: trace_durations.py <dir>"""
import csv
import glob
import sys

import numpy as np

files = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
d = {}
for f in files:
    for r in csv.DictReader(open(f)):
        key = (r["Kernel_Name"][:48], r.get("Grid_Size", r.get("Grid_Size_X", "")))
        d.setdefault(key, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in d.items():
    v = np.array(v)
    print("%-50s grid %-8s n=%5d min %.2f p10 %.2f med %.2f mean %.2f p90 %.2f us" %
          (k[0], k[1], len(v), v.min(), np.percentile(v, 10), np.median(v), v.mean(), np.percentile(v, 90)))
