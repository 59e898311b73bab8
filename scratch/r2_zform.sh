#!/bin/bash
# 17..128 rows: z-form of the 16-row kernel vs the MFMA-tiled (prefill) kernel, per shape
cd $GRAFT_REPO_ROOT
for M in 64 128; do
  for sh in wi logits qkv wo; do
    echo "z-form f32   : $(timeout -k 10 120 python scratch/kbench.py --shape $sh --M $M --f32 1 --graph 1 2>&1 | tail -1)"
    echo "tile kernel  : $(DIA_TUNE=gemm_mz_max=1,tile_min_blocks=1 timeout -k 10 120 python scratch/kbench.py --shape $sh --M $M --graph 1 2>&1 | tail -1)"
  done
done
