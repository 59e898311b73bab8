#!/bin/bash
# small GEMMs against the number of valid rows (rows >= M alias the last valid row: less L2 -> CU traffic for A)
for sh in o qkv; do for M in 2 4 5 8 12 16; do
  echo -n "$sh M=$M: "; python scratch/kbench.py --shape $sh --M $M 2>/dev/null | tail -1 | cut -d: -f2 | cut -c1-24
done; done
