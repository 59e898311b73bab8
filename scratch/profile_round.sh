#!/bin/bash
# regenerates the per-round evidence under gpurun_out/prof (copy what is judged into profiles/)
#   scratch/profile_round.sh r02
set -e
#   scratch/profile_round.sh r02 pmc      (only the counter passes + traffic.json)
R=${1:-r03}
O=$GRAFT_REPO_ROOT/gpurun_out/prof; mkdir -p $O
cd $GRAFT_REPO_ROOT
if [ "$2" != "pmc" ]; then
python bench.py 2>/dev/null | tail -1 > $O/${R}_bench_default.json; echo "bench default (batch 1 + configs) done"
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/${R}_bench_driver_style.json; echo "bench driver-style done"
python bench.py --batch 8 --cpu-steps 0 2>/dev/null | tail -1 > $O/${R}_bench_batch8.json; echo "bench b8 done"
python bench.py --batch 8 --pruned 0.5 --cpu-steps 0 2>/dev/null | tail -1 > $O/${R}_bench_batch8_pruned50.json; echo "bench b8 pruned done"
python bench.py --batch 1 --pruned 0.5 --cpu-steps 0 2>/dev/null | tail -1 > $O/${R}_bench_batch1_pruned50.json; echo "bench b1 pruned done"
python bench.py --kv f32 --cpu-steps 0 2>/dev/null | tail -1 > $O/${R}_bench_batch1_f32kv.json; echo "bench b1 f32 K/V done"
cd /tmp && export TMPDIR=/tmp
for cfg in "1:" "8:--batch 8" "8p:--batch 8 --pruned 0.5"; do
  tag=${cfg%%:*}; args=${cfg#*:}
  rm -rf /tmp/rp_$tag
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_$tag -- python3 $GRAFT_REPO_ROOT/bench.py $args --cpu-steps 0 --no-configs 2>/dev/null | tail -1 > $O/${R}_bench_batch${tag}_under_rocprof.json
  cp /tmp/rp_$tag/*/*_kernel_stats.csv $O/${R}_kernel_stats_batch$tag.csv; echo "rocprof $tag done"
done
fi
cd /tmp && export TMPDIR=/tmp
# PMC_PREHEAT: round 2 ran these passes with --preheat 0 because rocprofv3 --pmc died with SIGSEGV when the profiled process
# built a second decode session; since round 3 the session is drained before its graph is destroyed (DecodeSession.close,
# dia_engine_destroy) and the passes run with the default preheat unless PMC_PREHEAT says otherwise
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_$c -- python3 $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 2 --cpu-steps 0 --profile-steps 0 --no-configs ${PMC_PREHEAT:+--preheat $PMC_PREHEAT} > $O/${R}_pmc_$c.log 2>&1 || { echo "pmc $c pass FAILED (see ${R}_pmc_$c.log)"; tail -5 $O/${R}_pmc_$c.log; continue; }
  python3 - <<PY
import csv, glob, collections
f = glob.glob("/tmp/pmc_$c/*/*counter_collection.csv")[0]
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] != "$c": continue
    k = r["Kernel_Name"][:120]
    acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
with open("$O/${R}_pmc_$c.txt", "w") as o:
    for k, (s, n) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
        o.write(f"{k}\t launches {n}\t mean $c {s / n:.1f}\n")
PY
  echo "pmc $c done"
done
# HBM bytes per launch of every decode kernel (FETCH_SIZE counts KiB and reports 1/2 of a wide streaming read on gfx950:
# doubled; WRITE_SIZE exact), keyed the way bench.py names kernels -> profiles/traffic.json
python3 - <<PY
import json, re
def load(c):
    d = {}
    for line in open("$O/${R}_pmc_%s.txt" % c):
        name, rest = line.split("\t", 1)
        m = re.search(r"(k_[a-z0-9_]+(<[^>]*>)?)", name.replace("(anonymous namespace)::", ""))
        if not m or not name.lstrip("void ").replace("(anonymous namespace)::", "").startswith("k_"): continue
        d[m.group(1)] = float(rest.split("mean %s" % c)[1])
    return d
F, W = load("FETCH_SIZE"), load("WRITE_SIZE")
out = {"batch1": {}}
for k in F:
    out["batch1"][k] = {"hbm_bytes_per_launch": int(F[k] * 1024 * 2 + W.get(k, 0.0) * 1024), "FETCH_SIZE_KiB": F[k], "WRITE_SIZE_KiB": W.get(k, 0.0),
                        "source": "profiles/${R}_pmc_FETCH_SIZE.txt + ${R}_pmc_WRITE_SIZE.txt: rocprofv3 --pmc, separate passes over python bench.py --steps 8 --warmup 2 "
                                  "--cpu-steps 0 --profile-steps 0 --no-configs; FETCH_SIZE x 2 (gfx950 streaming-read correction) + WRITE_SIZE, mean per launch"}
json.dump(out, open("$O/traffic.json", "w"), indent=1)
print("traffic.json:", {k: v["hbm_bytes_per_launch"] for k, v in out["batch1"].items()})
PY
