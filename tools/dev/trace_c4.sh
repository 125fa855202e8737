#!/bin/bash
# dev: per-kernel totals of the c4 bench (64 x 2048^2 through BatchPlan) and how much of the wall time they cover
R=$(pwd); OUT=/tmp/tc4; rm -rf $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT -o kt --output-format csv -- python $R/bench.py --config c4 --no-extras --steps 3 --warmup 1 2>/dev/null | tail -1 | cut -c1-300
python - <<PY
import csv, glob, collections
rows = list(csv.DictReader(open(glob.glob("/tmp/tc4/*kernel_trace.csv")[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last step = last quarter of the launches
n = len(rows) // 4
rows = rows[-n:]
t0 = min(int(r["Start_Timestamp"]) for r in rows); t1 = max(int(r["End_Timestamp"]) for r in rows)
tot = collections.Counter(); cnt = collections.Counter()
for r in rows:
    k = r["Kernel_Name"].replace("siftk::", "").split("(")[0][:46]
    tot[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); cnt[k] += 1
print("span %.2f ms, %d launches, kernel time sum %.2f ms" % ((t1 - t0) / 1e6, len(rows), sum(tot.values()) / 1e6))
for k, v in tot.most_common(14):
    print("%-48s %5d  %9.1f us  avg %7.1f" % (k, cnt[k], v / 1e3, v / 1e3 / cnt[k]))
# busy time: union of kernel intervals
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows)
busy = 0; cs, ce = iv[0]
for s, e in iv[1:]:
    if s > ce: busy += ce - cs; cs, ce = s, e
    else: ce = max(ce, e)
busy += ce - cs
print("GPU busy (union of kernels) %.2f ms of %.2f" % (busy / 1e6, (t1 - t0) / 1e6))
PY
