#!/bin/bash
# 17..32-row kernels on the decode shapes at 32 rows: previous (k_gemm / k_gemm32), paired k_gemm16, k_gemm_blk32 forms
for sh in o qkv wi wo logits; do
  echo "== $sh"
  echo -n "previous:   "; DIA_DBG_PAIR16=0 DIA_DBG_BLK32=0 python scratch/kbench.py --shape $sh --M 32 --lend 1 2>/dev/null | tail -1
  echo -n "paired:     "; python scratch/kbench.py --shape $sh --M 32 --lend 1 $( [ $sh = wo ] && echo --sk 4 ) 2>/dev/null | tail -1
  echo -n "16 rows:    "; python scratch/kbench.py --shape $sh --M 16 $( [ $sh = wo ] && echo --sk 4 ) 2>/dev/null | tail -1
  if [ "$1" = all ]; then for f in 16,2 16,1 8,2 8,1; do
    echo -n "blk32 $f: "; DIA_DBG_PAIR16=0 DIA_DBG_BLK32=$f python scratch/kbench.py --shape $sh --M 32 --lend 1 2>/dev/null | tail -1
  done; fi
done
