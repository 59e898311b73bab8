#!/bin/bash
# z-form weight loads: temporal (default build) against non-temporal (scratch/libdia_nt.so, -DDIA_Z_TEMPORAL=0)
line() { python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('%-14s batch %2d  %8.1f frames/s  %.4f ms/step' % ('$1', d['config']['batch_per_gpu'], d['value'], d['ms_per_step']))"; }
for b in 9 16 32; do
  python bench.py --batch $b --steps 512 --cpu-steps 0 --profile-steps 0 2>/dev/null | tail -1 | line "temporal"
  DIA_HIP_LIB=$GRAFT_REPO_ROOT/scratch/libdia_nt.so python bench.py --batch $b --steps 512 --cpu-steps 0 --profile-steps 0 2>/dev/null | tail -1 | line "non-temporal"
done
