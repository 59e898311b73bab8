#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rp_pf
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_pf -- python3 $GRAFT_REPO_ROOT/scratch/prefill_time.py ${PF_BATCH:-1} > $O/r3_pf_prof.log 2>&1
cp /tmp/rp_pf/*/*_kernel_stats.csv $O/r3_prefill_kernel_stats.csv
python3 - <<'PY'
import csv, os
f = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r3_prefill_kernel_stats.csv")
rows = list(csv.DictReader(open(f)))
for r in rows[:22]:
    print(f'{r["Name"][:95]:95s} calls {r["Calls"]:>6s} avg {float(r["AverageNs"])/1e3:8.2f} us  total {float(r["TotalDurationNs"])/1e6:8.3f} ms')
PY
python3 - <<'PY'
import csv, os
f = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r3_prefill_kernel_stats.csv")
print("--- own kernels")
for r in csv.DictReader(open(f)):
    n = r["Name"].replace("(anonymous namespace)::", "")
    if "k_" in n and "at::" not in n:
        print(f'{n[:84]:84s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:8.2f} us  total {float(r["TotalDurationNs"])/1e6:7.3f} ms')
PY
