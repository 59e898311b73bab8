#!/bin/bash
# ring z-form (k_gemm16_zr) for 17..128 rows: kernel + engine parity, then per-op A/B against the one-strip-ahead z-form
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "gemm or act_f32" > $O/r3_zr_t1.log 2>&1 || { tail -25 $O/r3_zr_t1.log; exit 1; }
tail -2 $O/r3_zr_t1.log
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -s -k "large_batches or batch64 or pruned_compacted_batched" > $O/r3_zr_t2.log 2>&1 || { tail -25 $O/r3_zr_t2.log; exit 1; }
tail -3 $O/r3_zr_t2.log
for i in 1; do
  BATCHES="16 32 64" bash scratch/r3_big.sh
  echo "--- gemm_zr=0"
  DIA_TUNE=gemm_zr=0 BATCHES="16 32 64" bash scratch/r3_big.sh
done
echo "--- batch 1 / 8 (unconditional prefetch in every persistent form)"
BATCHES="1 8" bash scratch/r3_big.sh
