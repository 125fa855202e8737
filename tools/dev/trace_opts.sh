#!/bin/bash
# dev: kernel timeline of one keypoints() call with plan options:  bash tools/dev/trace_opts.sh "fork=1" [size] [octaves] [kind]
OPTS=${1:-base=1}; SIZE=${2:-4096}; OCT=${3:-3}; KIND=${4:-white}
R=$(pwd); OUT=$R/gpurun_out/trace_opts; rm -rf $OUT; mkdir -p $OUT
cat > $OUT/run.py <<PY
import sys, numpy as np, torch
sys.path.insert(0, "$R"); sys.path.insert(0, "$R/tests")
import sift_pyocl_amd as sp
from util import smooth_noise
img = smooth_noise(($SIZE, $SIZE)) if "$KIND" == "smooth" else np.random.default_rng(0).random(($SIZE, $SIZE), dtype=np.float32)
t = torch.from_numpy(img).cuda()
plan = sp.SiftPlan(shape=img.shape, dtype=np.float32, octave_max=$OCT or None)
for kv in "$OPTS".split(","):
    if "=" in kv and kv != "base=1":
        n, v = kv.split("="); plan.set_option(n, int(v))
for _ in range(8): k = plan.keypoints(t)
print(len(k), file=sys.stderr)
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT -o kt --output-format csv -- python $OUT/run.py > /dev/null 2> $OUT/err.txt
cd $R
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/trace_opts/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = max(i for i, r in enumerate(rows) if "minmax" in r["Kernel_Name"])
t0 = int(rows[idx]["Start_Timestamp"])
for r in rows[idx:]:
    s = int(r["Start_Timestamp"]) - t0; e = int(r["End_Timestamp"]) - t0
    name = r["Kernel_Name"].replace("siftk::", "").split("(")[0][:48]
    print("%9.1f %9.1f  dur %8.1f  q%-3s %s" % (s / 1e3, e / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), name))
PY
