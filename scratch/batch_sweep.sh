#!/bin/bash
# frames/s over the batch size (1024 steps each), written to gpurun_out/prof/r01_batch_sweep.txt
O=$GRAFT_REPO_ROOT/gpurun_out/prof; mkdir -p $O
: > $O/r01_batch_sweep.txt
for b in 1 2 4 8 12 16; do
  python bench.py --batch $b --cpu-steps 0 --profile-steps 0 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('batch %2d  %8.1f frames/s  %.4f ms/step  step-level %.3f of 8 TB/s' % ($b, d['value'], d['ms_per_step'], d['step_roofline']['frac']))" >> $O/r01_batch_sweep.txt
done
python scratch/prompt_bench.py 2>/dev/null | grep "B=" > $O/r01_prompt_prefill.txt
cat $O/r01_batch_sweep.txt $O/r01_prompt_prefill.txt
