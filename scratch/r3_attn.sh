#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "attn" > $O/r3_at_t1.log 2>&1 || { tail -25 $O/r3_at_t1.log; exit 1; }
tail -2 $O/r3_at_t1.log
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "bf16 or long_horizon or batch64 or large_batches or full_size_batch8" > $O/r3_at_t2.log 2>&1 || { tail -25 $O/r3_at_t2.log; exit 1; }
tail -2 $O/r3_at_t2.log
BATCHES="1 8 16 32 64" bash scratch/r3_big.sh
