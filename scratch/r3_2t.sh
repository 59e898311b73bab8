#!/bin/bash
# two-m-tile GEMM (k_gemm2t): kernel + engine parity, then per-op A/B against the z-form at 16 / 32 / 64 utterances
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "gemm or act_f32" > $O/r3_2t_t1.log 2>&1 || { tail -30 $O/r3_2t_t1.log; exit 1; }
tail -1 $O/r3_2t_t1.log
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -s -k "large_batches or batch64" > $O/r3_2t_t2.log 2>&1 || { tail -30 $O/r3_2t_t2.log; exit 1; }
tail -2 $O/r3_2t_t2.log
BATCHES="9 16 24 32 48 64" bash scratch/r3_big.sh
echo "--- gemm_2t=0"
DIA_TUNE=gemm_2t=0 BATCHES="9 16 32 64" bash scratch/r3_big.sh
