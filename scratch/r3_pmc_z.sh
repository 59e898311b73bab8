#!/bin/bash
# HBM-side traffic and L2 hit rate of the z-form GEMMs at 64 utterances per GPU (128 rows): are the 8 m-tile workgroups sharing their weights through L2?
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $c | cut -d' ' -f1)
  rm -rf /tmp/pmc_z_$tag
  DIA_TUNE=${ZTUNE:-gemm_zr=0} rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_z_$tag -- python3 $GRAFT_REPO_ROOT/bench.py --batch ${ZB:-64} --steps 4 --warmup 2 --cpu-steps 0 --profile-steps 0 --no-configs --preheat 0 > $O/r3_pmc_z_$tag.log 2>&1 || { tail -3 $O/r3_pmc_z_$tag.log; continue; }
  python3 - <<PY
import csv, glob, collections
f = glob.glob("/tmp/pmc_z_$tag/*/*counter_collection.csv")[0]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "")[:60]
    if "k_gemm16" not in k and "k_attn" not in k: continue
    k += " grid=" + r.get("Grid_Size", "?")
    a = acc[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k, d in sorted(acc.items()):
    print(k, {c: (round(v[0] / v[1], 1), v[1]) for c, v in d.items()})
PY
done
