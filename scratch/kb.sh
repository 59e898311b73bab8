#!/bin/bash
# usage: kb.sh "<shape> <nw> <M>" ...   — rocprofv3 kernel durations per configuration
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for cfg in "$@"; do
  set -- $cfg
  rm -rf /tmp/kb_prof
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kb_prof -- python scratch/kbench.py --shape $1 --nw $2 --M ${3:-2} --prefetch ${4:-0} --same ${5:-0} --sk ${6:-0} > /tmp/kb.log 2>&1
  grep "us/launch" /tmp/kb.log
  python - <<PY
import csv,glob
f=glob.glob("/tmp/kb_prof/*/*_kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "k_gem" in r["Name"] or "prefetch" in r["Name"]:
        print("   ", r["Name"][:60], r["Calls"], "calls avg %.2f us min %.2f" % (float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3))
PY
done
