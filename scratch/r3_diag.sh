#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -x -q -k "diagonal" > $O/r3_dg_t1.log 2>&1 || { tail -30 $O/r3_dg_t1.log; exit 1; }
tail -1 $O/r3_dg_t1.log
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -s -k "full_size_first or full_size_bf16_kv or free_running_tokens_mid or batched_equals_single" > $O/r3_dg_t2.log 2>&1 || { tail -30 $O/r3_dg_t2.log; exit 1; }
grep -E "K/V|passed|teacher" $O/r3_dg_t2.log | tail -6
one() { python bench.py $2 --cpu-steps 0 --no-configs 2>$O/r3_dg.err | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('%-26s %8.1f frames/s  %.4f ms/step ' % ('$1', d['value'], d['ms_per_step']), {k: round(v,2) for k,v in d['us_per_launch_by_op'].items()})"; }
for i in 1 2; do
  one "batch 1 wo diagonal" ""
  DIA_TUNE=wo_diag=0 one "batch 1 wo split-K 2" ""
done
one "batch 2 wo diagonal" "--batch 2"
DIA_TUNE=wo_diag=0 one "batch 2 wo split-K 2" "--batch 2"
