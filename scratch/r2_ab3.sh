#!/bin/bash
# per-op table: fp32 activation tiles (default from 5 rows on) vs three planes (act_f32=0) (DecodeSession reads the knob as on / off: there is no third setting)
cd $GRAFT_REPO_ROOT
one() { python bench.py $2 --cpu-steps 0 --no-configs 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('%-34s %8.1f frames/s ' % ('$1', d['value']), {k: round(v,2) for k,v in d['us_per_launch_by_op'].items()})"; }
for i in 1 2; do
  one "batch 8 fp32 tiles" "--batch 8"
  DIA_TUNE=act_f32=0 one "batch 8 planes" "--batch 8"
done
one "pruned batch 8 fp32 tiles" "--batch 8 --pruned 0.5"
DIA_TUNE=act_f32=0 one "pruned batch 8 planes" "--batch 8 --pruned 0.5"
one "batch 8 fp32 K/V fp32 tiles" "--batch 8 --kv f32"
DIA_TUNE=act_f32=0 one "batch 8 fp32 K/V planes" "--batch 8 --kv f32"
