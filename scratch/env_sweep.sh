#!/bin/bash
# runtime-environment A/B of the decode step (graph replay): kernarg placement, graph packet capture, ...
# usage: scratch/env_sweep.sh [bench args]      -> gpurun_out/env_sweep.txt
O=$GRAFT_REPO_ROOT/gpurun_out/env_sweep.txt; : > $O
run() { echo -n "$1 :: " >> $O; env $1 python bench.py --cpu-steps 0 --no-configs --profile-steps 0 "${@:2}" 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('%.1f frames/s  %.4f ms/step' % (d['value'], d['ms_per_step']))" >> $O 2>&1; }
for e in X=0 HIP_FORCE_DEV_KERNARG=1 HIP_FORCE_DEV_KERNARG=0 DEBUG_CLR_GRAPH_PACKET_CAPTURE=1 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 DEBUG_HIP_KERNARG_COPY_OPT=1 DEBUG_HIP_KERNARG_COPY_OPT=0 ROC_USE_FGS_KERNARG=0 ROC_USE_FGS_KERNARG=1 DEBUG_CLR_KERNARG_HDP_FLUSH_WA=0 GPU_MAX_HW_QUEUES=1 AMD_DIRECT_DISPATCH=0 ; do run $e "$@"; done
cat $O
