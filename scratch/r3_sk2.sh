#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/r3_sk2_tests.log 2>&1 || { tail -40 $O/r3_sk2_tests.log; exit 1; }
tail -2 $O/r3_sk2_tests.log
for rep in 1 2; do
echo "== strip-pair hand-off (rep $rep)"; BATCHES="12 16 24 32 48 64" bash scratch/r3_big.sh | grep "^batch" | cut -c1-260
echo "== gemm_2t=6: one hand-off per strip (rep $rep)"; DIA_TUNE=gemm_2t=6 BATCHES="12 16 24 32 48 64" bash scratch/r3_big.sh | grep "^batch" | cut -c1-260
done
