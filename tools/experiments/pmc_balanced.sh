#!/bin/bash
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/pmc_balanced.txt
export TMPDIR=/tmp
: > "$OUT"
i=0
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES" "GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS"; do
    i=$((i+1)); d=/tmp/prof_bal_$i; rm -rf $d
    ( cd /tmp && rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $d -- python "$ROOT/tools/experiments/pmc_balanced.py" > /dev/null 2>> /tmp/pmc_bal.err ) || echo "pass '$pass' failed" >> "$OUT"
    echo "# pass: $pass" >> "$OUT"
    python "$ROOT/tools/pmc_summary.py" $d >> "$OUT" 2>&1
done
tail -5 /tmp/pmc_bal.err >> "$OUT"
