#!/bin/bash
# strips per workgroup (DIA_DBG_SPW) of the k_gemm16 launches: paired 32-row form and the 16-row form
for sh in o qkv wi wo logits; do
  for M in 32 16; do
    for spw in 1 2 3 4 6 8 16; do
      echo -n "$sh M=$M spw=$spw: "; DIA_DBG_SPW=$spw python scratch/kbench.py --shape $sh --M $M --lend 1 $( [ $sh = wo ] && echo --sk 4 ) 2>/dev/null | tail -1 | cut -d: -f2 | cut -c1-20
    done
  done
done
