#!/bin/bash
# dev: HBM traffic (FETCH_SIZE, WRITE_SIZE; separate passes) of kernels matching $1 for each option set that follows
#   bash tools/dev/pmc_opt.sh blur_team_kernel "xcd_map=0" "xcd_map=1" [-- size=4096 octaves=3]
R=$(pwd); K=$1; shift
SETS=(); EXTRA=""
while [ $# -gt 0 ]; do if [ "$1" == "--" ]; then shift; EXTRA="$*"; break; fi; SETS+=("$1"); shift; done
cd /tmp && export TMPDIR=/tmp
for s in "${SETS[@]}"; do
  for c in FETCH_SIZE WRITE_SIZE; do
    OUT=$R/gpurun_out/pmc_opt/${s//[=,]/_}/$c; rm -rf $OUT; mkdir -p $OUT
    rocprofv3 --pmc $c --kernel-trace -d $OUT -o pmc --output-format csv -- python $R/tools/dev/run_opt.py "$s" $EXTRA > /dev/null 2> $OUT.err
  done
done
cd $R
python - "$K" <<'PY'
import csv, glob, collections, sys
K = sys.argv[1]
for d in sorted(glob.glob("gpurun_out/pmc_opt/*")):
    if not d.endswith(".err") and "/" in d:
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for f in glob.glob(d + "/*/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if K in r["Kernel_Name"]:
                    agg[(r["Kernel_Name"][:60], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for g in sorted(agg):
            print(d.split("/")[-1], g, {c: "%.1f" % (sum(v) / len(v)) for c, v in agg[g].items()}, "n=%d" % len(next(iter(agg[g].values()))))
PY
