#!/bin/bash
cd $GRAFT_REPO_ROOT
echo "== default"; python scratch/g2t_micro.py 128 64
for v in NOEPI NOMFMA NOWLOAD; do echo "== $v"; DIA_HIP_LIB=$GRAFT_REPO_ROOT/scratch/ab/lib_$v.so python scratch/g2t_micro.py 128; done
echo "== gemm_2t=3 (512 threads, KPW 4)"; DIA_TUNE=gemm_2t=3 python scratch/g2t_micro.py 128
