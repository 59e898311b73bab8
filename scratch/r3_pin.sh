#!/bin/bash
# prefetch pinned in front of the MFMAs in k_gemv_small / k_gemm16 (lib_pin.so) against the scheduler's placement (behind them), same box, interleaved
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for lib in new pin; do
  if [ $lib = new ]; then unset DIA_HIP_LIB; else export DIA_HIP_LIB=$GRAFT_REPO_ROOT/scratch/ab/lib_$lib.so; fi
  echo "== $lib (rep $rep)"
  BATCHES="${AB_BATCHES:-1 2 8 32}" bash scratch/r3_big.sh | grep "^batch"
done
done
