#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
python bench.py 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['metric'], d['value'], d['unit'], d['roofline']['frac'], d['cpu_baseline']['value'], d['prefill']['gpu_s']); print({k:v['frames_per_s'] for k,v in d['configs'].items()})"
