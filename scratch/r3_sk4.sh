#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -x -q -k "not long_horizon" > $O/r3_sk4_tests.log 2>&1 || { tail -40 $O/r3_sk4_tests.log; exit 1; }
tail -2 $O/r3_sk4_tests.log
for rep in 1 2; do
echo "== groups of four strips (rep $rep)"; BATCHES="24 32 48 64" bash scratch/r3_big.sh | grep "^batch" | cut -c1-260
echo "== gemm_2t=7: pairs only (rep $rep)"; DIA_TUNE=gemm_2t=7 BATCHES="24 32 48 64" bash scratch/r3_big.sh | grep "^batch" | cut -c1-260
done
