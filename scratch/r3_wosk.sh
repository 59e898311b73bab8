#!/bin/bash
cd $GRAFT_REPO_ROOT
for t in "" "wo_sk=1" "wo_sk=2" "wo_sk=4" "wo_pair=0"; do
  echo "== ${t:-default}"; DIA_TUNE=$t BATCHES="1 8" bash scratch/r3_big.sh | grep "^batch" | cut -c1-250
done
