#!/bin/bash
# regenerates the per-round evidence under gpurun_out/prof (copy what is judged into profiles/)
set -e
R=${1:-r01}
O=$GRAFT_REPO_ROOT/gpurun_out/prof; mkdir -p $O
cd $GRAFT_REPO_ROOT
python bench.py 2>/dev/null | tail -1 > $O/${R}_bench_batch1.json; echo "bench b1 done"
python bench.py --batch 8 --cpu-steps 0 2>/dev/null | tail -1 > $O/${R}_bench_batch8.json; echo "bench b8 done"
python bench.py --batch 8 --pruned 0.5 --cpu-steps 0 2>/dev/null | tail -1 > $O/${R}_bench_batch8_pruned50.json; echo "bench b8 pruned done"
python bench.py --batch 1 --pruned 0.5 --cpu-steps 0 2>/dev/null | tail -1 > $O/${R}_bench_batch1_pruned50.json; echo "bench b1 pruned done"
cd /tmp && export TMPDIR=/tmp
for b in 1 8; do
  rm -rf /tmp/rp_$b
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_$b -- python $GRAFT_REPO_ROOT/bench.py --batch $b --cpu-steps 0 2>/dev/null | tail -1 > $O/${R}_bench_batch${b}_under_rocprof.json
  cp /tmp/rp_$b/*/*_kernel_stats.csv $O/${R}_kernel_stats_batch$b.csv; echo "rocprof b$b done"
done
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_$c -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 2 --cpu-steps 0 --profile-steps 0 > /dev/null 2>&1
  python - <<PY
import csv, glob, collections
f = glob.glob("/tmp/pmc_$c/*/*counter_collection.csv")[0]
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] != "$c": continue
    k = r["Kernel_Name"][:90]
    acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
with open("$O/${R}_pmc_$c.txt", "w") as o:
    for k, (s, n) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
        o.write(f"{k}\t launches {n}\t mean $c {s / n:.1f}\n")
PY
  echo "pmc $c done"
done
