#!/bin/bash
# dev: kernel timeline of one keypoints() call on a small frame: bash tools/dev/trace_small.sh 512
R=$(pwd); S=${1:-512}; OUT=/tmp/ts; rm -rf $OUT
cat > /tmp/ts_run.py <<PY
import sys, os
sys.path.insert(0, "$R"); sys.path.insert(0, "$R/tests")
import numpy as np, torch, time
import sift_pyocl_amd as sp
img = np.random.default_rng(0).random(($S, $S), dtype=np.float32)
t = torch.from_numpy(img).cuda()
plan = sp.SiftPlan(shape=img.shape, dtype=np.float32, octave_max=int("${2:-0}") or None)
for _ in range(5): k = plan.keypoints(t)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): k = plan.keypoints(t)
print("call %.1f us, %d kp" % (1e6 * (time.perf_counter() - t0) / 20, len(k)))
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT -o kt --output-format csv -- python /tmp/ts_run.py 2>/dev/null | grep call
python - <<PY
import csv, glob
rows = list(csv.DictReader(open(glob.glob("/tmp/ts/*kernel_trace.csv")[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = max(i for i, r in enumerate(rows) if "minmax" in r["Kernel_Name"])   # (round 4: no begin_image launch any more)
t0 = int(rows[idx]["Start_Timestamp"])
for r in rows[idx:]:
    s = int(r["Start_Timestamp"]) - t0; e = int(r["End_Timestamp"]) - t0
    print("%8.1f %8.1f dur %6.1f q%-2s %s" % (s / 1e3, e / 1e3, (e - s) / 1e3, r["Queue_Id"], r["Kernel_Name"].replace("siftk::", "").split("(")[0][:44]))
PY
