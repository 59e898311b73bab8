#!/bin/bash
for valu in 0 1; do for b in 1 8; do
DIA_ATTN_VALU=$valu python bench.py --cpu-steps 0 --batch $b 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('valu $valu batch $b', d['value'], d['ms_per_step'], d['launch_breakdown']['attn_self'], d['launch_breakdown']['attn_cross'])"
done; done
