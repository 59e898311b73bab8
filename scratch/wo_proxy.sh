#!/bin/bash
# what bounds the wo GEMV (K 8192, N 2048) at M <= 4: per-CU traffic or the split-K seam?
r() { python scratch/kbench.py "$@" 2>/dev/null | tail -1 | cut -c1-80; }
echo -n "wo sk=1 (128 WG x 354 KB):      "; r --shape wo --M 2
echo -n "wo sk=2 (256 WG x 177 KB+seam): "; r --shape wo --M 2 --sk 2
echo -n "wo sk=4 (512 WG):               "; r --shape wo --M 2 --sk 4
echo -n "K8192 N4096 (256 WG x 354 KB):  "; r --K 8192 --N 4096 --M 2
echo -n "K4096 N4096 (256 WG x 177 KB):  "; r --K 4096 --N 4096 --M 2
echo -n "K4096 N2048 (128 WG x 177 KB):  "; r --K 4096 --N 2048 --M 2
echo -n "K2048 N8192 (512 WG x 88 KB):   "; r --K 2048 --N 8192 --M 2
echo -n "K2048 N4096 (256 WG x 88 KB):   "; r --K 2048 --N 4096 --M 2
echo -n "wi  (256 WG x 4 strips):        "; r --shape wi --M 2
