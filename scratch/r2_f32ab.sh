#!/bin/bash
# fp32 activation tiles vs three planes, per kernel shape at 16 rows (graph-replayed chains of 144 launches)
cd $GRAFT_REPO_ROOT
r() { timeout -k 10 120 python scratch/kbench.py "$@" --graph 1 2>/dev/null | tail -1; }
for sh in o qkv wi; do for f in 0 1; do echo "f32=$f $(r --shape $sh --M 16 --f32 $f)"; done; done
for f in 0 1; do echo "f32=$f $(r --shape wo --M 16 --sk 4 --f32 $f)"; done
echo "proxy (third plane re-reads the second):"
for sh in o wi; do echo "$(DIA_HIP_LIB=scratch/libdia_aproxy.so r --shape $sh --M 16)"; done
