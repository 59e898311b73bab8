#!/bin/bash
# batch 9-16 step: wo kernel choice A/B, then the batch sweep
O=$GRAFT_REPO_ROOT/gpurun_out/s2; mkdir -p $O
line() { python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('%-28s batch %2d  %8.1f frames/s  %.4f ms/step' % ('$1', d['config']['batch_per_gpu'], d['value'], d['ms_per_step']))"; }
python bench.py --batch 16 --steps 512 --cpu-steps 0 --profile-steps 0 2>/dev/null | tail -1 | line "wo paired"
DIA_DBG_WO_PAIR=0 python bench.py --batch 16 --steps 512 --cpu-steps 0 --profile-steps 0 2>/dev/null | tail -1 | line "wo k_gemm32"
DIA_DBG_WO_PAIR=0 DIA_DBG_PAIR16=0 python bench.py --batch 16 --steps 512 --cpu-steps 0 --profile-steps 0 2>/dev/null | tail -1 | line "previous kernels"
for b in 8 9 12; do python bench.py --batch $b --steps 512 --cpu-steps 0 --profile-steps 0 2>/dev/null | tail -1 | line "default"; done
python bench.py --batch 8 --pruned 0.5 --steps 512 --cpu-steps 0 --profile-steps 0 2>/dev/null | tail -1 | line "pruned 0.5"
python bench.py --batch 16 --pruned 0.5 --steps 512 --cpu-steps 0 --profile-steps 0 2>/dev/null | tail -1 | line "pruned 0.5"
