#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/prof; mkdir -p $O
python scratch/prefill_time.py 2>/dev/null | grep "pass" | tee $O/r03_prefill_time.txt
echo "--- knobs off: gemm_2t=5 (run-time tails), ckv_merge=0 (one cross-K/V launch per layer)" | tee -a $O/r03_prefill_time.txt
DIA_TUNE=gemm_2t=5,ckv_merge=0 python scratch/prefill_time.py 1 2>/dev/null | grep "pass [23]" | tee -a $O/r03_prefill_time.txt
bash scratch/r3_prefill_prof.sh | grep -A14 "own kernels" | tee $O/r03_prefill_kernels.txt
