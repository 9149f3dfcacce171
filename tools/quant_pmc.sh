#!/bin/bash
# SQ / GRBM counters of the quantiser's kernels (two --pmc passes, kernel-trace only), mean per dispatch
# usage (GPU box): bash tools/quant_pmc.sh [K N] > gpurun_out/quant_pmc.txt
ROOT=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/qpmc1 /tmp/qpmc2
timeout 150 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU \
    --kernel-trace --output-format csv -d /tmp/qpmc1 -- python "$ROOT/tools/quant_one.py" "$@" > /tmp/qpmc1.log 2>&1 || tail -3 /tmp/qpmc1.log
timeout 150 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR \
    --kernel-trace --output-format csv -d /tmp/qpmc2 -- python "$ROOT/tools/quant_one.py" "$@" > /tmp/qpmc2.log 2>&1 || tail -3 /tmp/qpmc2.log
python "$ROOT/tools/pmc_summary.py" /tmp/qpmc1 2>&1 | grep -v "at::native\|^   .*n=  *[0-9]* mean *0.0$" | grep -A9 "quant_pack\|colmax\|strip_quant" 
python "$ROOT/tools/pmc_summary.py" /tmp/qpmc2 2>&1 | grep -A8 "quant_pack\|colmax\|strip_quant"
