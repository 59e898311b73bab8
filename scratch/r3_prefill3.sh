#!/bin/bash
# merged cross-K/V launch: tests, then the prefill's GPU time with one launch per layer beside it
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -x -q > $O/r3_pf3_t1.log 2>&1 || { tail -25 $O/r3_pf3_t1.log; exit 1; }
tail -2 $O/r3_pf3_t1.log
echo "== default (merged)"; python scratch/prefill_time.py 2>/dev/null | grep "pass [23]"
echo "== ckv_merge=0"; DIA_TUNE=ckv_merge=0 python scratch/prefill_time.py 2>/dev/null | grep "pass [23]"
bash scratch/r3_prefill_prof.sh | grep -A12 "own kernels"
