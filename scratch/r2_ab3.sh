#!/bin/bash
# batch 8 per-op table: fp32 tiles (default) / planes with the fp32 code present / planes with it compiled out
cd $GRAFT_REPO_ROOT
one() { python bench.py --batch 8 --cpu-steps 0 --no-configs 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('%-28s %8.1f frames/s ' % ('$1', d['value']), {k: round(v,2) for k,v in d['us_per_launch_by_op'].items()})"; }
for i in 1 2; do
  one "fp32 tiles"
  DIA_TUNE=act_f32=0 one "planes (flag off)"
  DIA_TUNE=act_f32=0 DIA_HIP_LIB=scratch/libdia_noact.so one "planes (compiled out)"
done
