# Round-3 experiment script (run on the GPU box from the repo root); output under profiles/ -- see profiles/README.md
run() { echo "== $*"; env "$@" ./tools/chain_probe 64 alloc 2>&1 | grep -E "^A0|40 separate 16 MiB allocations, rotated" | head -2; }
run X=1
run HIP_FORCE_DEV_KERNARG=1
run HIP_FORCE_DEV_KERNARG=0
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run HIP_FORCE_DEV_KERNARG=1 DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run GPU_MAX_HW_QUEUES=1
run HSA_XNACK=0
run ROC_SIGNAL_POOL_SIZE=128
