#!/bin/bash
line() { python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('%-22s batch %2d  %8.1f frames/s  %.4f ms/step' % ('$1', d['config']['batch_per_gpu'], d['value'], d['ms_per_step']))"; }
for b in 4 8; do
  python bench.py --batch $b --cpu-steps 0 --profile-steps 0 2>/dev/null | tail -1 | line "default"
  DIA_DBG_SPW=1 python bench.py --batch $b --cpu-steps 0 --profile-steps 0 2>/dev/null | tail -1 | line "spw=1 everywhere"
done
python bench.py --batch 16 --cpu-steps 0 --profile-steps 0 2>/dev/null | tail -1 | line "default"
