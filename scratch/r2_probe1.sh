#!/bin/bash
# round-2 probes: skeleton sizes, split-K on the small GEMMs, sampler phase stamps
cd $GRAFT_REPO_ROOT
timeout -k 10 120 scratch/launchcost/launchcost || exit 1
r() { timeout -k 10 120 python scratch/kbench.py "$@" --graph 1 2>/dev/null | tail -1; }
for M in 2 16; do for sk in 0 2; do echo "o M=$M sk=$sk: $(r --shape o --M $M --sk $sk)"; done; done
for sk in 0 2; do echo "qkv M=16 sk=$sk: $(r --shape qkv --M 16 --sk $sk)"; done
echo "--- sampler stamps, Dia-1.6B"
DIA_HIP_LIB=scratch/libdia_dbg.so timeout -k 10 200 python scratch/sstamps.py full 2>/dev/null | tail -8
