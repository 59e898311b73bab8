#!/bin/bash
cd $GRAFT_REPO_ROOT
for b in 16 32; do
  python bench.py --batch $b --steps 256 --cpu-steps 0 --no-configs 2>/dev/null | tail -1 > gpurun_out/r2_b$b.json
  python - <<PY
import json
d=json.load(open("gpurun_out/r2_b$b.json"))
print("batch $b:", d["value"], "frames/s", d["ms_per_step"], "ms/step")
for r in d["roofline_by_kernel"]:
    print("  %-44s x%3d  %7.2f us  share %.3f  frac %.3f  %s" % (r["kernel"][:44], r["launches_per_step"], r["us_per_launch"], r["share_of_step_time"], r.get("frac_of_hbm_peak", 0) or 0, ",".join(r.get("ops", []))[:40]))
PY
done
