#!/bin/bash
# rocprofv3 --pmc passes (separate runs, kernel-trace only) over tools/experiments/pmc_mid_vs_splitk.py:
# HBM / L2 traffic and SQ counters of the medium-batch kernels.  usage: tools/pmc_medium.sh <out.txt>
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${1:-$ROOT/gpurun_out/pmc_medium.txt}
export TMPDIR=/tmp
: > "$OUT"
i=0
for pass in "FETCH_SIZE" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM"; do
    i=$((i+1)); d=/tmp/prof_medium_$i; rm -rf $d
    ( cd /tmp && rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $d -- python "$ROOT/tools/experiments/pmc_mid_vs_splitk.py" > /dev/null 2>> /tmp/pmc_medium.err ) || echo "pass '$pass' failed" >> "$OUT"
    echo "# pass: $pass" >> "$OUT"
    python "$ROOT/tools/pmc_summary.py" $d >> "$OUT" 2>&1
done
