#!/bin/bash
# dev: host API calls and kernels of ONE keypoints() call on one time axis (where does the GPU wait for the host?):
#   bash tools/dev/trace_host.sh [size] [octaves] [opts]
SIZE=${1:-4096}; OCT=${2:-3}; OPTS=${3:-base=1}
R=$(pwd); OUT=$R/gpurun_out/trace_host; rm -rf $OUT; mkdir -p $OUT
cat > $OUT/run.py <<PY
import sys, numpy as np, torch
sys.path.insert(0, "$R")
import sift_pyocl_amd as sp
img = np.random.default_rng(0).random(($SIZE, $SIZE), dtype=np.float32)
t = torch.from_numpy(img).cuda()
plan = sp.SiftPlan(shape=img.shape, dtype=np.float32, octave_max=$OCT or None)
for kv in "$OPTS".split(","):
    if "=" in kv and kv != "base=1":
        n, v = kv.split("="); plan.set_option(n, int(v))
for _ in range(30): k = plan.keypoints(t)
print(len(k), file=sys.stderr)
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --hip-runtime-trace -d $OUT -o kt --output-format csv -- python $OUT/run.py > /dev/null 2> $OUT/err.txt
cd $R
ls $OUT
python - <<'PY'
import csv, glob
kf = glob.glob("gpurun_out/trace_host/*kernel_trace.csv")[0]
af = glob.glob("gpurun_out/trace_host/*hip_api_trace.csv")[0]
K = list(csv.DictReader(open(kf))); A = list(csv.DictReader(open(af)))
K.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = max(i for i, r in enumerate(K) if "minmax" in r["Kernel_Name"])
t0 = int(K[idx]["Start_Timestamp"])
ev = []
for r in K[idx:]:
    ev.append((int(r["Start_Timestamp"]) - t0, "K %8.1f  q%-2s %s" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Queue_Id", "?"), r["Kernel_Name"].replace("siftk::", "").split("(")[0][:44])))
for r in A:
    s = int(r["Start_Timestamp"]) - t0
    if s > -60000 and s < 900000:
        ev.append((s, "    host %6.1f us  %s" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Function"])))
ev.sort()
for s, txt in ev: print("%9.1f  %s" % (s / 1e3, txt))
PY
