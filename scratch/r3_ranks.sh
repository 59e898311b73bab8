#!/bin/bash
# rehearsal of the driver's N > 1 invocations on ONE GPU (gloo, all ranks on cuda:0): the default per-GPU batch of configs[4] at N = 4 and N = 2
cd $GRAFT_REPO_ROOT
export DIA_BENCH_SHARE_DEVICE=1 HSA_ENABLE_IPC_MODE_LEGACY=0
for n in 4 2; do
  timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500 + n)) bench.py --gpus $n --steps 64 --warmup 4 2>gpurun_out/r3_ranks_$n.err | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('N=%d' % d['n_gpus'], d['value'], d['unit'], d['scaling'], d['config']['parallelism'], d['config']['batch_per_gpu'], d['config']['workload'][:160])" || { tail -20 gpurun_out/r3_ranks_$n.err; exit 1; }
done
