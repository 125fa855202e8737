#!/bin/bash
# dev: average kernel durations (rocprofv3 --kernel-trace) of one plan's calls for several builds of the library
#   bash tools/dev/kt_libs.sh "minmax|blur_team" base mm8 [-- size=4096 octaves=3]
R=$(pwd); PAT=$1; shift
TAGS=(); EXTRA=""
while [ $# -gt 0 ]; do if [ "$1" == "--" ]; then shift; EXTRA="$*"; break; fi; TAGS+=("$1"); shift; done
cp sift_pyocl_amd/libsiftmi.so /tmp/libsiftmi_keep.so
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
for t in "${TAGS[@]}"; do
  cp $R/sift_pyocl_amd/libsiftmi_$t.so $R/sift_pyocl_amd/libsiftmi.so
  OUT=$R/gpurun_out/kt_libs/$t.$rep; rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --kernel-trace -d $OUT -o kt --output-format csv -- python $R/tools/dev/run_opt.py "xcd_map=1" n=24 $EXTRA > /dev/null 2> $OUT.err
  python - "$OUT" "$PAT" "$t.$rep" <<'PY'
import csv, glob, sys, re, collections
d, pat, tag = sys.argv[1:4]
agg = collections.defaultdict(list)
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if re.search(pat, r["Kernel_Name"]):
            agg[re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void siftk::", "")[:44]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k in sorted(agg):
    v = sorted(agg[k][len(agg[k]) // 3:])          # (skip the first calls: clocks ramping)
    print("%-8s %-46s n=%3d  median %7.2f us  min %7.2f" % (tag, k, len(v), v[len(v) // 2], v[0]))
PY
done
done
cp /tmp/libsiftmi_keep.so $R/sift_pyocl_amd/libsiftmi.so
