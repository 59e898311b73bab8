#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "split or paired or uniform or act_f32" > $O/r3_sk_t1.log 2>&1 || { tail -25 $O/r3_sk_t1.log; exit 1; }
tail -2 $O/r3_sk_t1.log
BATCHES="16 32 64" bash scratch/r3_big.sh | grep "^batch"
