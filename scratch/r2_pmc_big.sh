#!/bin/bash
# HBM fetch bytes per launch of the z-form GEMMs at batch 16 / 32: do the m-tile workgroups of a strip share its weights through L2?
cd /tmp && export TMPDIR=/tmp
for b in 8 32; do
  rm -rf /tmp/pmcb_$b
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmcb_$b -- python3 $GRAFT_REPO_ROOT/bench.py --batch $b --steps 8 --warmup 2 --cpu-steps 0 --profile-steps 0 --no-configs --preheat 0 > /dev/null 2>&1
  python3 - <<PY
import csv, glob, collections
f = glob.glob("/tmp/pmcb_$b/*/*counter_collection.csv")[0]
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] != "FETCH_SIZE": continue
    k = r["Kernel_Name"].replace("void (anonymous namespace)::", "")[:60]
    acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
print("batch $b")
for k, (s, n) in sorted(acc.items(), key=lambda kv: -kv[1][0])[:6]:
    print("   %-60s launches %6d  mean HBM fetch %.2f MB (FETCH_SIZE x 2 KiB)" % (k, n, s / n * 2 * 1024 / 1e6))
PY
done
