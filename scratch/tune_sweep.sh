#!/bin/bash
# A/B of launch-heuristic knobs (DIA_TUNE) on the whole decode step: scratch/tune_sweep.sh "<bench args>" knob=value,... knob=value,...
O=$GRAFT_REPO_ROOT/gpurun_out/tune_sweep.txt
args="$1"; shift
for t in "$@"; do
  echo -n "[$args] DIA_TUNE=$t :: " >> $O
  DIA_TUNE=$t python bench.py $args --cpu-steps 0 --no-configs 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); o=d['us_per_launch_by_op']; print('%.1f frames/s  %.4f ms/step  wo %.2f wi %.2f o %.2f qkv %.2f self %.2f cross %.2f' % (d['value'], d['ms_per_step'], o['wo'], o['wi'], o['o'], o['qkv'], o['attn_self'], o['attn_cross']))" >> $O 2>&1
done
