#!/bin/bash
# short-prompt prefill after the one-pass k_enc_kv_planes and the 256-thread k_gemm2t (K = 1024): tests, then GPU time with the old forms beside it
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -x -q > $O/r3_pf2_t1.log 2>&1 || { tail -25 $O/r3_pf2_t1.log; exit 1; }
tail -2 $O/r3_pf2_t1.log
echo "== default"; python scratch/prefill_time.py 2>/dev/null | grep "pass [23]"
echo "== gemm_2t=3 (512-thread K = 1024 form)"; DIA_TUNE=gemm_2t=3 python scratch/prefill_time.py 1 2>/dev/null | grep "pass [23]"
bash scratch/r3_prefill_prof.sh | grep -A12 "own kernels"
