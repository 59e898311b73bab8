#!/bin/bash
# persistent MLP segment: kernel test, full-size parity, A/B of the batch-1 step (seg on / off), per-op table
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_seg.py -x -q -s > $O/r3_seg_t1.log 2>&1; rc=$?
tail -15 $O/r3_seg_t1.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -s -k "full_size_first or full_size_bf16 or free_running_tokens_mid" > $O/r3_seg_t2.log 2>&1; rc=$?
tail -8 $O/r3_seg_t2.log
[ $rc -ne 0 ] && exit $rc
one() { python bench.py $2 --cpu-steps 0 --no-configs 2>$O/r3_seg_b.err | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('%-28s %8.1f frames/s  %.4f ms/step  launches %s ' % ('$1', d['value'], d['ms_per_step'], d.get('launches_per_step')), {k: round(v,2) for k,v in d['us_per_launch_by_op'].items()})"; }
for i in 1 2; do
  one "batch 1 segments" ""
  DIA_TUNE=seg=0 one "batch 1 launches" ""
done
one "batch 2 segments" "--batch 2"
DIA_TUNE=seg=0 one "batch 2 launches" "--batch 2"
one "batch 1 f32 K/V segments" "--kv f32"
