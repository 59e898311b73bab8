#!/bin/bash
# GEMM kernel tests + large-batch parity, then per-op tables at 1 / 8 / 16 / 32 / 64 utterances and the prefill passes
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "gemm or act_f32" > $O/r3_all_t1.log 2>&1 || { tail -25 $O/r3_all_t1.log; exit 1; }
tail -1 $O/r3_all_t1.log
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "large_batches or batch64 or pruned_compacted_batched or full_size_first or teacher_forced or encoder" > $O/r3_all_t2.log 2>&1 || { tail -25 $O/r3_all_t2.log; exit 1; }
tail -1 $O/r3_all_t2.log
BATCHES="1 8 16 32 64" bash scratch/r3_big.sh
python scratch/prefill_time.py 1 8 2>&1 | grep "pass 3"
