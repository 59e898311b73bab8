#!/bin/bash
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pf
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf -- python $GRAFT_REPO_ROOT/bench.py --batch ${1:-8} --steps 2 --warmup 1 --cpu-steps 0 --profile-steps 0 > /tmp/pf.log 2>&1
tail -1 /tmp/pf.log | cut -c1-300
python - <<PY
import csv,glob
f=glob.glob("/tmp/pf/*/*_kernel_stats.csv")[0]
rows=[r for r in csv.DictReader(open(f)) if "anonymous" in r["Name"] or "k_" in r["Name"][:12]]
for r in rows[:14]:
    print(r["Name"][:70].ljust(70), r["Calls"].rjust(6), "tot %.2f ms avg %.1f us" % (float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3))
PY
