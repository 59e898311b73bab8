#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -x -q -k "not long_horizon" > $O/r3_sk5_tests.log 2>&1 || { tail -40 $O/r3_sk5_tests.log; exit 1; }
tail -2 $O/r3_sk5_tests.log
echo "== default"; python scratch/prefill_time.py 1 2>/dev/null | grep "pass [23]"
echo "== gemm_2t=6 (one hand-off per strip)"; DIA_TUNE=gemm_2t=6 python scratch/prefill_time.py 1 2>/dev/null | grep "pass [23]"
