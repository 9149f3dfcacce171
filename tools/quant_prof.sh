#!/bin/bash
# per-kernel split of quant_weights under rocprofv3 (new column-major kernel, then the older row-major one)
# usage (GPU box): bash tools/quant_prof.sh > gpurun_out/quant_prof.txt
export EETQ_AMD_TUNING=1   # the EETQ_AMD_QUANT_* A/B hooks answer only with this switch (csrc/common.hpp: tuning_env)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp
cd /tmp
for mode in colmajor strip; do
    d=/tmp/qp_$mode; rm -rf $d
    if [ $mode = strip ]; then export EETQ_AMD_QUANT_KERNEL=strip; fi
    timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o q -- python "$ROOT/tools/quant_one.py" "$@" > /tmp/qp_$mode.log 2>&1
    f=$(find $d -name '*kernel_stats.csv' 2>/dev/null | head -1)
    echo "== $mode"
    if [ -n "$f" ]; then head -6 "$f" | cut -c1-200; else tail -5 /tmp/qp_$mode.log; fi
done
