#!/bin/bash
# dev: extra PMC counters (instruction fetch, scalar memory, branches) of bench.py kernels matching a name ($1)
R=$(pwd); K=${1:-orientation_kernel}; OUT=$R/gpurun_out/pmc2_$K; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH SQ_WAVE_CYCLES" "SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_INST_CYCLES_SMEM SQ_INSTS_SALU" "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_INST_CYCLES_VMEM_RD SQ_WAIT_ANY" "SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace -d $OUT/g$i -o pmc --output-format csv -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extras > /dev/null 2> $OUT/g$i.err
done
cd $R
python - "$K" <<'PY'
import csv, glob, collections, sys
K = sys.argv[1]
for f in sorted(glob.glob("gpurun_out/pmc2_%s/g*/*counter_collection.csv" % K)):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    nd = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        if K in r["Kernel_Name"]:
            agg[r["Grid_Size"]][r["Counter_Name"]] += float(r["Counter_Value"])
            nd[r["Grid_Size"]].add(r["Dispatch_Id"])
    for g in sorted(agg):
        print("grid", g, "dispatches", len(nd[g]), {c: round(v / len(nd[g])) for c, v in agg[g].items()})
PY
