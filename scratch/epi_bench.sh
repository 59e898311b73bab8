#!/bin/bash
line() { python -c "
import sys,json; d=json.loads(sys.stdin.read()); b=d['launch_breakdown'] or {}; print('%-8s batch %2d  %8.1f frames/s  %.4f ms/step  o %s co %s wi %s wo %s' % ('$1', d['config']['batch_per_gpu'], d['value'], d['ms_per_step'], b.get('o'), b.get('co'), b.get('wi'), b.get('wo')))"; }
for b in "$@"; do
  DIA_HIP_LIB=$GRAFT_REPO_ROOT/scratch/libdia_old.so python bench.py --batch $b --cpu-steps 0 2>/dev/null | tail -1 | line old
  python bench.py --batch $b --cpu-steps 0 2>/dev/null | tail -1 | line new
done
