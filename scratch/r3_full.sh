#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/r3_full_tests.log 2>&1 || { tail -40 $O/r3_full_tests.log; exit 1; }
tail -3 $O/r3_full_tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" 2>&1 | tail -3
