#!/bin/bash
# dev: PMC counters of the standalone blur harness (one counter group per pass)
R=$(pwd); OUT=$R/gpurun_out/pmc_blur; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE SQ_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_SMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace -d $OUT/g$i -o pmc --output-format csv -- $R/tools/ubench/blur_abl_0 > /dev/null 2> $OUT/g$i.err
done
cd $R
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pmc_blur/g*/*counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        n = "N27" if "Li27E" in k else ("N15" if "Li15E" in k else ("N11" if "Li11E" in k else k[:20]))
        agg[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for n in sorted(agg):
        print(n, {c: round(sum(v) / len(v)) for c, v in agg[n].items()})
PY
