#!/bin/bash
# SQ counters of the int4 vs int8 M = 1 GEMV (separate --pmc passes, kernel-trace only).  usage: bash tools/experiments/int4_gemv_pmc.sh [K N]
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
export TMPDIR=/tmp
for i in 1 2 3; do rm -rf /tmp/prof_i4_$i; done
( cd /tmp && rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT \
    --kernel-trace --output-format csv -d /tmp/prof_i4_1 -- python "$ROOT/tools/experiments/int4_gemv_pmc.py" "$@" > /dev/null 2>&1 )
( cd /tmp && rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU \
    --kernel-trace --output-format csv -d /tmp/prof_i4_2 -- python "$ROOT/tools/experiments/int4_gemv_pmc.py" "$@" > /dev/null 2>&1 )
( cd /tmp && rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAVES SQ_LEVEL_WAVES \
    --kernel-trace --output-format csv -d /tmp/prof_i4_3 -- python "$ROOT/tools/experiments/int4_gemv_pmc.py" "$@" > /dev/null 2>&1 )
for i in 1 2 3; do python "$ROOT/tools/pmc_summary.py" /tmp/prof_i4_$i 2>&1 | grep -v "^$"; done
