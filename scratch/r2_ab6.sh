#!/bin/bash
# batch 1 / 2 (M <= 4): fp32 activation tiles (default) vs three planes (act_f32=0), same library
cd $GRAFT_REPO_ROOT
one() { python bench.py $2 --cpu-steps 0 --no-configs 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('%-30s %8.1f frames/s ' % ('$1', d['value']), {k: round(v,2) for k,v in d['us_per_launch_by_op'].items()})"; }
for i in 1 2 3; do
  one "batch 1 fp32 tiles" ""
  DIA_TUNE=act_f32=0 one "batch 1 planes" ""
done
one "batch 2 fp32 tiles" "--batch 2"
DIA_TUNE=act_f32=0 one "batch 2 planes" "--batch 2"
one "pruned batch 1 fp32 tiles" "--pruned 0.5"
DIA_TUNE=act_f32=0 one "pruned batch 1 planes" "--pruned 0.5"
one "batch 1 fp32 K/V fp32 tiles" "--kv f32"
DIA_TUNE=act_f32=0 one "batch 1 fp32 K/V planes" "--kv f32"
