#!/bin/bash
# round 3, first GPU call: GPU tests (new: batch 64 at mid + full size, 1024-step horizon), default bench line, batch sweep 1..64
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O/prof
timeout -k 10 900 python -m pytest tests -m gpu -x -q -s > $O/r3_tests.log 2>&1; rc=$?
tail -5 $O/r3_tests.log
grep -E "long horizon|batch 64|max-abs" $O/r3_tests.log | tail -30
[ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python bench.py > $O/prof/r03_bench_default_a.json 2> $O/r3_bench.err || { tail -5 $O/r3_bench.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open("gpurun_out/prof/r03_bench_default_a.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"])
for k,v in d["configs"].items(): print(k, v["frames_per_s"], v["ms_per_step"], v["prefill_gpu_ms"])
print(d["cpu_baseline"])
PY
ROUND=r03 timeout -k 10 600 scratch/batch_sweep.sh
