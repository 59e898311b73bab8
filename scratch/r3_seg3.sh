#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
for cfg in "seg_dbg=1,seg_sleep=0" "seg_dbg=1,seg_sleep=64"; do
  echo "=== $cfg"
  DIA_TUNE=$cfg timeout -k 10 120 python scratch/seg_stamps.py 2 2>&1 | grep -v amdgpu.ids | tail -26
done > $O/seg_stamps_c.txt 2>&1
cat $O/seg_stamps_c.txt
