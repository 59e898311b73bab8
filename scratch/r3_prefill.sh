#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -x -q -k "enc or crosskv or encoder or edge_texts or prompt or dia_api or teacher_forced" > $O/r3_pf_t1.log 2>&1 || { tail -25 $O/r3_pf_t1.log; exit 1; }
tail -2 $O/r3_pf_t1.log
python scratch/prefill_time.py
