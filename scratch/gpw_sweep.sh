#!/bin/bash
# granules per wave before the keys of a (row, head) are split over another workgroup: self (DIA_DBG_GPW) and
# cross (DIA_DBG_GPW_CROSS) attention, whole-step effect at batch 1 and 8
line() { python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('%-26s batch %2d  %8.1f frames/s  %.4f ms/step' % ('$1', d['config']['batch_per_gpu'], d['value'], d['ms_per_step']))"; }
for b in 1 8; do
  python bench.py --batch $b --cpu-steps 0 --profile-steps 0 2>/dev/null | tail -1 | line "default (1 / 1)"
  for g in 2 3 4; do DIA_DBG_GPW=$g python bench.py --batch $b --cpu-steps 0 --profile-steps 0 2>/dev/null | tail -1 | line "self gpw=$g"; done
  for g in 2 4; do DIA_DBG_GPW_CROSS=$g python bench.py --batch $b --cpu-steps 0 --profile-steps 0 2>/dev/null | tail -1 | line "cross gpw=$g"; done
done
