#!/bin/bash
# LDS conflict counters of descriptor_kernel for the current lib
R=$(pwd); OUT=/tmp/pmc_lds; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS --kernel-trace -d $OUT -o pmc --output-format csv -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extras --no-steady > /dev/null 2> $OUT/err
cd $R
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/pmc_lds/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "descriptor_kernel" in r["Kernel_Name"]:
            agg[r["Grid_Size"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for g in agg:
    print("grid", g, {c: (round(sum(v[0::2]) / max(1, len(v[0::2]))), round(sum(v[1::2]) / max(1, len(v[1::2])))) for c, v in agg[g].items()})
PY
