#!/bin/bash
cd $GRAFT_REPO_ROOT
echo "== default"; BATCHES="1 2 4 8" bash scratch/r3_big.sh | grep "^batch" | cut -c1-250
echo "== no split-K hand-off in k_gemv_small / k_gemm16 (timing only, wrong results)"; DIA_HIP_LIB=$GRAFT_REPO_ROOT/scratch/ab/lib_noho2.so BATCHES="1 2 4 8" bash scratch/r3_big.sh | grep "^batch" | cut -c1-250
