#!/bin/bash
# dev: PMC counters of bench.py kernels matching a name ($1); launches of the two keypoint groups reported separately
# (group 0 = even, group 1 = odd dispatches of the same grid)
R=$(pwd); K=${1:-descriptor_kernel}; OUT=$R/gpurun_out/pmcs_$K; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VMEM_RD" "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32" "SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM" "GRBM_GUI_ACTIVE SQ_INSTS_BRANCH SQ_IFETCH SQ_IFETCH_LEVEL"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace -d $OUT/g$i -o pmc --output-format csv -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extras > /dev/null 2> $OUT/g$i.err
done
cd $R
python - "$K" <<'PY'
import csv, glob, collections, sys
K = sys.argv[1]
for f in sorted(glob.glob("gpurun_out/pmcs_%s/g*/*counter_collection.csv" % K)):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if K in r["Kernel_Name"]:
            agg[r["Grid_Size"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for g in sorted(agg):
        print("grid", g, {c: (round(sum(v[0::2]) / max(1, len(v[0::2]))), round(sum(v[1::2]) / max(1, len(v[1::2])))) for c, v in agg[g].items()})
PY
rm -rf $OUT
