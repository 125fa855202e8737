#!/bin/bash
# dev: PMC counters of bench.py kernels whose name contains $1, averaged per (kernel name, grid), one counter group per pass
R=$(pwd); K=${1:-blur_team_kernel}; OUT=$R/gpurun_out/pmcn_$K; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA" "SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace -d $OUT/g$i -o pmc --output-format csv -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-steady > /dev/null 2> $OUT/g$i.err
done
cd $R
python - "$K" <<'PY'
import csv, glob, collections, sys, re
K = sys.argv[1]
tab = collections.defaultdict(dict)
for f in sorted(glob.glob("gpurun_out/pmcn_%s/g*/*counter_collection.csv" % K)):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if K in r["Kernel_Name"]:
            m = re.search(r"<[^>]*>", r["Kernel_Name"])
            agg[(m.group(0) if m else r["Kernel_Name"][:40], int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for key, cs in agg.items():
        for c, v in cs.items():
            tab[key][c] = round(sum(v) / len(v))
            tab[key]["n"] = len(v)
for key in sorted(tab, key=lambda k: (-k[1], k[0])):
    print(key[0], "grid", key[1], tab[key])
PY
rm -rf $OUT
