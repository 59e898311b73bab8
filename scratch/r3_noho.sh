#!/bin/bash
cd $GRAFT_REPO_ROOT
echo "== default"; BATCHES="16 32 64" bash scratch/r3_big.sh | grep "^batch"
echo "== no split-K hand-off in k_gemm2t (timing only, wrong results)"; DIA_HIP_LIB=$GRAFT_REPO_ROOT/scratch/ab/lib_noho.so BATCHES="16 32 64" bash scratch/r3_big.sh | grep "^batch"
