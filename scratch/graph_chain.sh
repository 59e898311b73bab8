#!/bin/bash
# per-launch time of ONE kernel repeated in a replayed graph (144 launches over 18 cold weight buffers): what the kernel
# costs when nothing else shares the instruction cache, against its average inside the decode step
r() { python scratch/kbench.py "$@" --graph 1 2>/dev/null | tail -1; }
for M in 2 16; do for sh in o qkv wi wo logits; do r --shape $sh --M $M $( [ $sh = wo ] && echo --sk $( [ $M = 2 ] && echo 2 || echo 4 ) ); done; done
echo "same buffer every launch (L2 / Infinity Cache resident):"; r --shape o --M 2 --same 1; r --shape wi --M 2 --same 1
