#!/bin/bash
# uniform tails (compile-time epilogue, unconditional stores, pinned prefetch) in k_gemm2t: tests, then 16..64 rows with the run-time tail beside it (gemm_2t=5)
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -x -q > $O/r3_uni_t1.log 2>&1 || { tail -25 $O/r3_uni_t1.log; exit 1; }
tail -2 $O/r3_uni_t1.log
for rep in 1 2; do
echo "== uniform (rep $rep)"; BATCHES="16 32 64" bash scratch/r3_big.sh | grep "^batch"
echo "== gemm_2t=5 run-time tail (rep $rep)"; DIA_TUNE=gemm_2t=5 BATCHES="16 32 64" bash scratch/r3_big.sh | grep "^batch"
done
python scratch/prefill_time.py 1 2>/dev/null | grep "pass [23]"
