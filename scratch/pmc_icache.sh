#!/bin/bash
# instruction-cache behaviour of the decode kernels inside the step (kernels alternate, 70 KB of code per layer
# against a 64 KB instruction cache per CU pair): per-kernel means per launch
O=$GRAFT_REPO_ROOT/gpurun_out/s2; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_ic
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES --output-format csv -d /tmp/pmc_ic -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 2 --cpu-steps 0 --profile-steps 0 > /dev/null 2>&1
python - <<PY
import csv, glob, collections
f = glob.glob("/tmp/pmc_ic/*/*counter_collection.csv")[0]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:80]
    if "anonymous" not in k: continue
    a = acc[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
with open("$O/r01_pmc_icache.txt", "w") as o:
    for k, d in sorted(acc.items()):
        n = max(v[1] for v in d.values())
        o.write(f"{k}\t launches {n}\t" + "\t".join(f"{c} {v[0] / v[1]:.0f}" for c, v in sorted(d.items())) + "\n")
print(open("$O/r01_pmc_icache.txt").read())
PY
