#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_seg.py -x -q > $O/r3_seg_t1.log 2>&1 || { tail -15 $O/r3_seg_t1.log; exit 1; }
tail -2 $O/r3_seg_t1.log
for nb in 3 2 1; do for sl in 0 4 12; do
  echo "=== seg_nb=$nb seg_sleep=$sl"
  DIA_TUNE=seg_nb=$nb,seg_sleep=$sl timeout -k 10 120 python scratch/seg_stamps.py 2 2>&1 | grep -v amdgpu.ids | tail -18
done; done > $O/seg_stamps_b.txt 2>&1
grep -E "===|eager|15 end|co computed|wi computed" $O/seg_stamps_b.txt
