#!/bin/bash
# dev: PMC counters of bench.py kernels matching a name ($1), one counter group per pass
R=$(pwd); K=${1:-descriptor_kernel}; OUT=$R/gpurun_out/pmc_$K; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA" "SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LDS_ATOMIC_RETURN SQ_LDS_ADDR_CONFLICT" "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_FLAT SQ_INSTS_LDS"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace -d $OUT/g$i -o pmc --output-format csv -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline > /dev/null 2> $OUT/g$i.err
done
cd $R
python - "$K" <<'PY'
import csv, glob, collections, sys
K = sys.argv[1]
for f in sorted(glob.glob("gpurun_out/pmc_%s/g*/*counter_collection.csv" % K)):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if K in r["Kernel_Name"]:
            agg[r["Grid_Size"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for g in sorted(agg):
        print("grid", g, {c: round(sum(v) / len(v)) for c, v in agg[g].items()}, "n=%d" % len(next(iter(agg[g].values()))))
PY
