#!/bin/bash
# all-thread CROSSKV tail (k_gemm2t<8, false, false, 4, true>): tests, then the prefill with the shared tail beside it (gemm_2t=4)
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -x -q > $O/r3_pf5_t1.log 2>&1 || { tail -25 $O/r3_pf5_t1.log; exit 1; }
tail -2 $O/r3_pf5_t1.log
echo "== default"; python scratch/prefill_time.py 2>/dev/null | grep "pass [23]"
echo "== gemm_2t=4 (shared CROSSKV tail)"; DIA_TUNE=gemm_2t=4 python scratch/prefill_time.py 1 2>/dev/null | grep "pass [23]"
bash scratch/r3_prefill_prof.sh | grep -A12 "own kernels"
