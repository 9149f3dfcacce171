#!/bin/bash
# rocprofv3 --kernel-trace --stats of the HIP-graph decode of config 5 (Llama-2-13B shapes, prompt 1024, batch 1):
# per-kernel call counts and average durations of the decode step, our kernels and the rest.
# usage (GPU box, repo root): tools/profile_decode.sh [tag] [extra llama_generate.py flags]  ->  gpurun_out/<tag>_llama13b_decode_kernel_stats.txt
set -u
TAG=${1:-rXX}
shift || true
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
rm -rf /tmp/prof_decode
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_decode -- python "$ROOT/examples/llama_generate.py" \
    --accelerate --graph --new 200 "$@" > "$OUT/${TAG}_decode_under_rocprof.json" 2> "$OUT/${TAG}_decode_rocprof.err" ) || echo "rocprofv3 run failed" >&2
STATS=$(find /tmp/prof_decode -name '*kernel_stats.csv' | head -1)
python - "$STATS" "$*" > "$OUT/${TAG}_llama13b_decode_kernel_stats.txt" <<'PYEOF'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print("# rocprofv3 --kernel-trace --stats -- python examples/llama_generate.py --accelerate --graph --new 200 %s" % sys.argv[2])
print("# 40 layers x ~205 decoded/warm-up tokens per per-layer kernel; prefill and model build are in the tail")
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:24]:
    print("%-74s calls %6d avg %8.2f us total %8.1f ms" % (r["Name"][:74], int(r["Calls"]), float(r["AverageNs"]) / 1e3,
                                                           float(r["TotalDurationNs"]) / 1e6))
PYEOF
