#!/bin/bash
# after removing the waits in front of the first weight request (DIA_OPAQUE16 etc.): kernel tests, parity, then per-op times
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -m gpu -x -q > $O/r3_early_tests.log 2>&1 || { tail -30 $O/r3_early_tests.log; exit 1; }
tail -3 $O/r3_early_tests.log
BATCHES="1 2 8 16 32 64" bash scratch/r3_big.sh
