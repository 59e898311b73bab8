#!/bin/bash
mkdir -p gpurun_out/s2
DIA_HIP_LIB=$GRAFT_REPO_ROOT/scratch/libdia_old.so python scratch/epi_ab.py /tmp/old.npz 2>/dev/null | tail -1
python scratch/epi_ab.py /tmp/new.npz 2>/dev/null | tail -1
python - <<PY
import numpy as np
a, b = np.load("/tmp/old.npz"), np.load("/tmp/new.npz")
bad = [k for k in a.files if not np.array_equal(a[k].view(np.uint8), b[k].view(np.uint8))]
print("arrays", len(a.files), "differing", bad)
for k in bad[:6]:
    x, y = a[k].astype(np.float64), b[k].astype(np.float64); i = np.argwhere(x != y)
    print(k, "n diff", len(i), "first", i[0], x[tuple(i[0])], y[tuple(i[0])])
PY
