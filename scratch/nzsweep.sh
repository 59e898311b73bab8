#!/bin/bash
for nz in 0 1 2 8; do for b in 1 8; do
DIA_DBG_NZ=$nz python bench.py --cpu-steps 0 --batch $b 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('nz $nz batch $b', d['value'], d['ms_per_step'], d['launch_breakdown']['attn_self'], d['launch_breakdown']['attn_cross'])"
done; done
