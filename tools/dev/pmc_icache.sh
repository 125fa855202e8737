#!/bin/bash
# dev: instruction-fetch counters of the kernels whose name contains $1 (one plan, headline frame)
R=$(pwd); K=${1:-blur_team_kernel}; OUT=$R/gpurun_out/pmc_icache; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -o "SQC_[A-Z_0-9]*\|SQ_IFETCH[A-Z_]*\|SQ_WAIT_IFETCH[A-Z_]*\|SQ_INST_LEVEL_[A-Z_]*" | sort -u | tr '\n' ' ' > $OUT/avail.txt
cat $OUT/avail.txt; echo
i=0
for grp in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_IFETCH SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace -d $OUT/g$i -o pmc --output-format csv -- python $R/tools/dev/run_opt.py "xcd_map=1" n=4 > /dev/null 2> $OUT/g$i.err
  tail -2 $OUT/g$i.err
done
cd $R
python - "$K" <<'PY'
import csv, glob, collections, sys, re
K = sys.argv[1]
tab = collections.defaultdict(dict)
for f in sorted(glob.glob("gpurun_out/pmc_icache/g*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if K in r["Kernel_Name"]:
            m = re.search(r"<[^>]*>", r["Kernel_Name"])
            agg[(m.group(0) if m else r["Kernel_Name"][:40], int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for key, cs in agg.items():
        for c, v in cs.items():
            tab[key][c] = round(sum(v) / len(v))
for key in sorted(tab, key=lambda k: (-k[1], k[0])):
    print(key[0], "grid", key[1], tab[key])
PY
