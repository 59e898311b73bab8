#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_LDS SQ_INSTS_VALU" "SQ_WAIT_INST_LDS SQ_WAVE_CYCLES" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
  rm -rf /tmp/pmc_t
  rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_t -- python scratch/kbench.py --shape ewi --M 1696 --reps 1 > /tmp/pmc_t.log 2>&1 || { echo "failed: $c"; tail -3 /tmp/pmc_t.log; continue; }
  python - <<PY
import csv, glob, collections
fs = glob.glob("/tmp/pmc_t/*/*counter_collection.csv")
if not fs: print("no csv for $c"); raise SystemExit
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(fs[0])):
    if "k_gemm_tile" not in r["Kernel_Name"]: continue
    acc[r["Counter_Name"]][0] += float(r["Counter_Value"]); acc[r["Counter_Name"]][1] += 1
for k, (s, n) in acc.items(): print(k, "mean per launch %.4g over %d" % (s / n, n))
PY
done
