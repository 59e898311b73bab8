#!/bin/bash
r() { python scratch/kbench.py "$@" 2>/dev/null | tail -1 | cut -c1-60; }
for M in 16 32 64; do for sk in 4 8 16; do echo -n "wo M=$M sk=$sk: "; r --shape wo --M $M --sk $sk; done; done
for M in 32 64; do for spw in 1 2 4 8; do echo -n "wo M=$M sk=8 spw=$spw: "; DIA_DBG_SPW=$spw python scratch/kbench.py --shape wo --M $M --sk 8 2>/dev/null | tail -1 | cut -c1-60; done; done
