"""Split the gemv dispatches of a `rocprofv3 --kernel-trace -- python bench.py` run by phase of bench.py.

bench.py issues the GEMV in this order: 1 parity call, W warm-up launches (eager), K graph-replayed launches (untimed
upload replay), K graph-replayed launches (the timed region), 20 eager, K eager launches carrying event pairs (the
roofline pass).  Graph-replayed launches run back to back (the next kernel's waves start while the previous one drains),
so their begin..end spans are longer than those of the event-timed eager launches the roofline fraction is quoted on.
usage: python tools/split_bench_trace.py <kernel_trace.csv> [steps] [warmup]
"""
import csv
import sys

import numpy as np


def main():
    path = sys.argv[1]
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    warm = int(sys.argv[3]) if len(sys.argv) > 3 else 200
    rows = [r for r in csv.DictReader(open(path)) if "gemv_kernel" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    d = np.array([(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows])
    segs = [("parity", 1), ("warm-up (eager)", warm), ("graph pre-capture warm (eager, side stream)", 3),
            ("graph replay (upload)", steps), ("graph replay (timed region)", steps), ("eager warm", 20),
            ("eager, event-timed (roofline pass)", steps)]
    i = 0
    print("gemv dispatches: %d, overall mean %.3f us" % (len(d), d.mean()))
    for name, n in segs:
        seg = d[i:i + n]
        if len(seg):
            print("%-48s n=%5d mean %.3f median %.3f min %.3f us" % (name, len(seg), seg.mean(), np.median(seg), seg.min()))
        i += n
    if i < len(d):
        print("%-48s n=%5d mean %.3f" % ("rest", len(d) - i, d[i:].mean()))


if __name__ == "__main__":
    main()
