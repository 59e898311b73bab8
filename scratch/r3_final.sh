#!/bin/bash
# end-of-round evidence: full GPU suite, smoke, the r03 profile set, the batch sweep, prefill timing
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O/prof
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/r3_final_tests.log 2>&1 || { tail -40 $O/r3_final_tests.log; exit 1; }
tail -2 $O/r3_final_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
PMC_PREHEAT=0 bash scratch/profile_round.sh r03 > $O/profile_round_r03.log 2>&1 || { tail -20 $O/profile_round_r03.log; exit 1; }
tail -4 $O/profile_round_r03.log
ROUND=r03 bash scratch/batch_sweep.sh
python scratch/prefill_time.py 2>/dev/null | grep "pass [123]" | tee $O/prof/r03_prefill_time.txt
