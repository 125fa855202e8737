#!/bin/bash
# dev (round 3): baseline + descriptor ablation + FP64 instruction split on one box
R=$(pwd); OUT=$R/gpurun_out/r3b1; rm -rf $OUT; mkdir -p $OUT
python bench.py --no-cpu-baseline > $OUT/bench_base.json 2> $OUT/bench_base.err
(cd /tmp && rocprofv3 -L > $OUT/counters.txt 2>&1)
grep -o "SQ_INSTS_VALU[A-Z0-9_]*" $OUT/counters.txt | sort -u > $OUT/valu_counters.txt
# ablation build of the library
cp sift_pyocl_amd/libsiftmi.so /tmp/libsiftmi_keep.so
cp gpurun_in_libsiftmi_ablate.so sift_pyocl_amd/libsiftmi.so
for a in 0 11 12 13 14 15 16; do echo "== ablate $a"; SIFTMI_ABLATE=$a python tools/stage_profile.py 4096 white 3 float32 overlap=0 2>&1 | grep -E "descriptors group|orientation_assignment group|TOTAL|keypoints"; done > $OUT/ablate_white.txt 2>&1
cp /tmp/libsiftmi_keep.so sift_pyocl_amd/libsiftmi.so
# FP64 / transcendental split of the descriptor and orientation kernels
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64" "SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32" "SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT" "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace -d $OUT/pmc/g$i -o pmc --output-format csv -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extras > /dev/null 2> $OUT/pmc_g$i.err
done
cd $R
python - <<'PY' > $OUT/pmc_fp64.txt 2>&1
import csv, glob, collections
for K in ("descriptor_kernel", "orientation_kernel"):
    print("==", K)
    for f in sorted(glob.glob("gpurun_out/r3b1/pmc/g*/*counter_collection.csv")):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            if K in r["Kernel_Name"]:
                agg[r["Grid_Size"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for g in sorted(agg):
            # launches alternate group 0 / group 1: report even and odd dispatches separately
            print("grid", g, {c: (round(sum(v[0::2]) / max(1, len(v[0::2]))), round(sum(v[1::2]) / max(1, len(v[1::2])))) for c, v in agg[g].items()})
PY
rm -rf $OUT/pmc
ls -la $OUT
