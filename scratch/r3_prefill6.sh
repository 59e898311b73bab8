#!/bin/bash
# planes all-thread tails with compile-time epilogues: tests, prefill time (knob 5 = shared tails), micro
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -x -q > $O/r3_pf6_t1.log 2>&1 || { tail -25 $O/r3_pf6_t1.log; exit 1; }
tail -2 $O/r3_pf6_t1.log
echo "== default"; python scratch/prefill_time.py 2>/dev/null | grep "pass [23]"
echo "== gemm_2t=5 (shared tails)"; DIA_TUNE=gemm_2t=5 python scratch/prefill_time.py 1 2>/dev/null | grep "pass [23]"
python scratch/g2t_micro.py 128 2>/dev/null
bash scratch/r3_prefill_prof.sh | grep -A14 "own kernels"
