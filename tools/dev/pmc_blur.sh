#!/bin/bash
# dev: stall / pipeline counters of blur_team_kernel per template instance and grid (one counter group per pass, --pmc with
# --kernel-trace only, as the pool requires).  bash tools/dev/pmc_blur.sh [kernel substring] > profiles/<round>/pmc_blur_team_kernel.txt
R=$(pwd); K=${1:-blur_team_kernel}; OUT=$R/gpurun_out/pmcb_$K; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VALU SQ_THREAD_CYCLES_VALU" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS" \
           "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INSTS_SALU" \
           "SQ_IFETCH SQ_IFETCH_LEVEL SQC_ICACHE_MISSES SQC_ICACHE_HITS" \
           "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_INSTS_BRANCH" \
           "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace -d $OUT/g$i -o pmc --output-format csv -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extras --no-steady > /dev/null 2> $OUT/g$i.err
done
cd $R
python - "$K" <<'PY'
import csv, glob, collections, sys, re
K = sys.argv[1]
tab = collections.defaultdict(dict)
for f in sorted(glob.glob("gpurun_out/pmcb_%s/g*/*counter_collection.csv" % K)):
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
    for r in csv.DictReader(open(f)):
        if K in r["Kernel_Name"]:
            m = re.search(r"<[^>]*>", r["Kernel_Name"])
            agg[(m.group(0) if m else r["Kernel_Name"][:40], int(r["Grid_Size"]))][r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    for key, cs in agg.items():
        for c, per in cs.items():
            v = list(per.values())
            tab[key][c] = round(sum(v) / len(v))
            tab[key]["n"] = len(v)
print("# per launch (summed over the chip's counter instances, averaged over the launches of 6 images): instance, grid threads, counters")
for key in sorted(tab, key=lambda k: (-k[1], k[0])):
    t = tab[key]
    print(key[0], "grid", key[1], {k: t[k] for k in sorted(t)})
    wc = t.get("SQ_WAVE_CYCLES", 0)
    if wc:
        def pct(c): return "%s %.1f%%" % (c, 100.0 * t.get(c, 0) / wc)
        print("    share of wave-cycles:", ", ".join(pct(c) for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_MISC")))
        la = t.get("SQ_LDS_IDX_ACTIVE", 0)
        if la: print("    LDS: bank-conflict cycles / active cycles = %.1f%%; busy cycles (per-XCD sum) %d" % (100.0 * t.get("SQ_LDS_BANK_CONFLICT", 0) / la, t.get("SQ_BUSY_CYCLES", 0)))
PY
rm -rf $OUT
