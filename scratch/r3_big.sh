#!/bin/bash
# per-op microseconds of the step at 16 / 32 / 64 utterances per GPU (the shards of BASELINE configs[4] at N = 4, 2, 1)
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
for b in ${BATCHES:-16 32 64}; do
  python bench.py --batch $b --steps 256 --cpu-steps 0 --no-configs 2>$O/r3_big.err | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('batch %2d  %8.1f frames/s  %.4f ms/step ' % ($b, d['value'], d['ms_per_step']), {k: round(v,2) for k,v in d['us_per_launch_by_op'].items()}); print('   ', [(r['kernel'][:48], r['launches_per_step'], r['us_per_launch'], r['frac']) for r in d['roofline_by_kernel'][:6]])"
done
