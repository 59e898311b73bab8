#!/bin/bash
cd $GRAFT_REPO_ROOT
echo "== default"; BATCHES="16 32 64" bash scratch/r3_big.sh | grep "^batch"
echo "== gemm_2t=2 (two m-tiles per workgroup wherever the shape allows)"; DIA_TUNE=gemm_2t=2 BATCHES="16 32 64" bash scratch/r3_big.sh | grep "^batch"
