#!/bin/bash
# Which kernels make up the config-5 leg (13B shapes: prefill of 1024 tokens once + 50 graph-decoded tokens)?
# usage (GPU box): bash tools/prefill_profile.sh > gpurun_out/prefill_kernels.txt
ROOT=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/pp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -o p -- python "$ROOT/bench.py" --no-cpu-baseline --steps 100 --warmup 10 --gemm-steps 20 > /tmp/pp.log 2>&1
f=$(find /tmp/pp -name '*kernel_stats.csv' 2>/dev/null | head -1)
if [ -n "$f" ]; then
    python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time %.1f ms" % (tot / 1e6))
for r in rows[:32]:
    print("%8.2f ms %6d calls avg %9.1f us  %s" % (float(r["TotalDurationNs"]) / 1e6, int(r["Calls"]), float(r["AverageNs"]) / 1e3, r["Name"][:110]))
PY
else
    tail -5 /tmp/pp.log
fi
