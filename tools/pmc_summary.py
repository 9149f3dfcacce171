"""Summarise rocprofv3 --pmc output (counter_collection csv) per kernel: mean of every counter per dispatch.

usage: python tools/pmc_summary.py <rocprof output dir> [name filter substring]
"""
import csv
import glob
import sys
from collections import defaultdict


def main():
    root = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    files = glob.glob(root + "/**/*counter_collection.csv", recursive=True)
    acc = defaultdict(lambda: defaultdict(list))
    for f in files:
        for row in csv.DictReader(open(f)):
            name = row.get("Kernel_Name", "")
            if flt and flt not in name:
                continue
            acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for name, ctrs in sorted(acc.items()):
        short = name[:110]
        print(short)
        for c, v in sorted(ctrs.items()):
            print("   %-28s n=%3d mean %16.1f" % (c, len(v), sum(v) / len(v)))
    traces = glob.glob(root + "/**/*kernel_trace.csv", recursive=True)
    dur = defaultdict(list)
    for f in traces:
        for row in csv.DictReader(open(f)):
            name = row.get("Kernel_Name", "")
            if flt and flt not in name:
                continue
            dur[name].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
    for name, v in sorted(dur.items()):
        v.sort()
        print("%-110s n=%3d dur med %8.2f us" % (name[:110], len(v), v[len(v) // 2]))


if __name__ == "__main__":
    main()
