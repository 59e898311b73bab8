#!/bin/bash
cd $GRAFT_REPO_ROOT
echo "== default"; BATCHES="4 8" bash scratch/r3_big.sh | grep "^batch" | cut -c1-250
echo "== k_gemm16 reads ONE k-tile of its activation image per wave (timing only, wrong results)"; DIA_HIP_LIB=$GRAFT_REPO_ROOT/scratch/ab/lib_noa.so BATCHES="4 8" bash scratch/r3_big.sh | grep "^batch" | cut -c1-250
