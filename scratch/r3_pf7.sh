#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_kernels.py -x -q -k "not long_horizon" > $O/r3_pf7_t1.log 2>&1 || { tail -25 $O/r3_pf7_t1.log; exit 1; }
tail -2 $O/r3_pf7_t1.log
python scratch/prefill_time.py 2>/dev/null | grep "pass [123]"
python bench.py --steps 64 --cpu-steps 0 --no-configs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['prefill'])"
python bench.py --batch 8 --steps 64 --cpu-steps 0 --no-configs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['prefill'])"
