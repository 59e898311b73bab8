#!/bin/bash
# workgroup budget of the planes k_gemm2t launches (A traffic = workgroups x 192..384 KB against zp x W)
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -x -q > $O/r3_pf4_t1.log 2>&1 || { tail -25 $O/r3_pf4_t1.log; exit 1; }
tail -2 $O/r3_pf4_t1.log
echo "== default"; python scratch/prefill_time.py 2>/dev/null | grep "pass [23]"
for wgs in 512 256 192 128 96 64; do
  echo "== g2t_wgs=$wgs"; DIA_TUNE=g2t_wgs=$wgs python scratch/prefill_time.py 1 2>/dev/null | grep "pass [23]"
done
bash scratch/r3_prefill_prof.sh | grep -A12 "own kernels"
