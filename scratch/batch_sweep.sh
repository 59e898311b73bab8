#!/bin/bash
# frames/s over the batch size (1024 steps each up to 16, 512 beyond), written to gpurun_out/prof/${ROUND:-r03}_batch_sweep.txt
R=${ROUND:-r03}
O=$GRAFT_REPO_ROOT/gpurun_out/prof; mkdir -p $O
cd $GRAFT_REPO_ROOT
[ -z "$SWEEP" ] && : > $O/${R}_batch_sweep.txt
for b in ${SWEEP:-1 2 4 8 9 12 16 24 32 48 64}; do
  st=1024; [ $b -gt 16 ] && st=512
  python bench.py --batch $b --steps $st --cpu-steps 0 --profile-steps 0 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('batch %2d  %8.1f frames/s  %.4f ms/step  step-level %.3f of 8 TB/s' % ($b, d['value'], d['ms_per_step'], d['step_roofline']['frac']))" >> $O/${R}_batch_sweep.txt
done
cat $O/${R}_batch_sweep.txt
