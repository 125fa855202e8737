#!/bin/bash
# dev: kernel trace of a few bench steps; prints the timeline of the last image (start/end relative to its first kernel)
R=$(pwd); OUT=$R/gpurun_out/trace_gaps; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT -o kt --output-format csv -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-pipelined > /dev/null 2> $OUT/err.txt
cd $R
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/trace_gaps/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last image = from the last min/max pass (the first kernel of an image)
idx = max(i for i, r in enumerate(rows) if "minmax" in r["Kernel_Name"])   # (round 4: no begin_image launch any more)
t0 = int(rows[idx]["Start_Timestamp"])
prev_end = t0
for r in rows[idx:]:
    s = int(r["Start_Timestamp"]) - t0; e = int(r["End_Timestamp"]) - t0
    name = r["Kernel_Name"].replace("siftk::", "").split("(")[0][:48]
    print("%9.1f %9.1f  dur %8.1f  q%-3s %s" % (s / 1e3, e / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), name))
PY
