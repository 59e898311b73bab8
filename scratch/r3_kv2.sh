#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "attn_self or crosskv" > $O/r3_kv2_t1.log 2>&1 || { tail -30 $O/r3_kv2_t1.log; exit 1; }
tail -1 $O/r3_kv2_t1.log
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -s -k "full_size_bf16_kv or long_horizon or batched_equals_single" > $O/r3_kv2_t2.log 2>&1 || { tail -30 $O/r3_kv2_t2.log; exit 1; }
grep -E "K/V|passed|long horizon" $O/r3_kv2_t2.log
for kv in bf16 bf16x2 f32; do for b in 1 8; do
  python bench.py --batch $b --kv $kv --cpu-steps 0 --no-configs 2>$O/r3_kv2.err | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('batch %d %-7s %8.1f frames/s  %.4f ms/step ' % ($b, '$kv', d['value'], d['ms_per_step']), {k: round(v,2) for k,v in d['us_per_launch_by_op'].items() if 'attn' in k})"
done; done
