#!/bin/bash
# batch 17-32 step: k_gemm16 over 3-4 m-tiles against the previous kernels (generic k_gemm / tile kernel)
line() { python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('%-22s batch %2d  %8.1f frames/s  %.4f ms/step' % ('$1', d['config']['batch_per_gpu'], d['value'], d['ms_per_step']))"; }
for b in 48 64; do
  python bench.py --batch $b --steps 256 --cpu-steps 0 --profile-steps 0 2>/dev/null | tail -1 | line "z = m-tiles"
  DIA_DBG_PAIR16=4 python bench.py --batch $b --steps 256 --cpu-steps 0 --profile-steps 0 2>/dev/null | tail -1 | line "previous kernels"
done
