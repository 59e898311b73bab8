#!/bin/bash
# self-attention at full length (1024 steps, batch 8 and 1): granules per wave / key splits
cd $GRAFT_REPO_ROOT
for t in "" "attn_gpw=2" "attn_nz=4" "attn_gpw=2,attn_nz=4" "attn_gpw=3" "attn_nz=16"; do
  for b in 8 1; do
    DIA_TUNE=$t python bench.py --batch $b --steps 1024 --cpu-steps 0 --no-configs 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); o=d['us_per_launch_by_op']; print('%-22s batch %d  %8.1f frames/s  attn_self %.2f (end of run)  attn_cross %.2f' % ('$t' or 'default', $b, d['value'], o['attn_self'], o['attn_cross']))"
  done
done
