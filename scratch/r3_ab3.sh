#!/bin/bash
# same box, three libraries: HEAD, new without DIA_OPAQUE16, new; twice each (interleaved) to see the noise
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for lib in head noopq new; do
  if [ $lib = new ]; then unset DIA_HIP_LIB; else export DIA_HIP_LIB=$GRAFT_REPO_ROOT/scratch/ab/lib_$lib.so; fi
  echo "== $lib (rep $rep)"
  BATCHES="${AB_BATCHES:-1 8 32}" bash scratch/r3_big.sh | grep "^batch"
done
done
