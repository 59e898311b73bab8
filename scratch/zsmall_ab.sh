#!/bin/bash
# 5..16 rows: 4-row kernel over row groups (DIA_DBG_ZSMALL = highest strips x groups; 0 = off) against k_gemm16
line() { python -c "
import sys,json; d=json.loads(sys.stdin.read()); b=d['launch_breakdown'] or {}; print('%-14s batch %2d  %8.1f frames/s  %.4f ms/step  qkv %s o %s cq %s co %s' % ('$1', d['config']['batch_per_gpu'], d['value'], d['ms_per_step'], b.get('qkv'), b.get('o'), b.get('cq'), b.get('co')))"; }
for b in 3 4 6 8; do
  for z in 0 512 768; do DIA_DBG_ZSMALL=$z python bench.py --batch $b --cpu-steps 0 2>/dev/null | tail -1 | line "zsmall=$z"; done
done
