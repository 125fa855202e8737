#!/bin/bash
# VALU instructions of ONE headline frame by kernel family and op class (rocprofv3 PMC, one counter group per pass; no
# trace domains besides --kernel-trace).  From the repo root on the GPU box:  bash tools/valu_frame.sh [outdir]
# -> <outdir>/valu_frame.json (read by bench.py's roofline_valu) and valu_frame.txt (the table).
R=$(pwd); OUT=${1:-$R/gpurun_out/valu_frame}; RAW=/tmp/valu_frame_raw; rm -rf $RAW; mkdir -p $OUT $RAW
cd /tmp && export TMPDIR=/tmp
STEPS=4; WARM=2
i=0
for grp in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32" \
           "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64" \
           "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_SALU" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace -d $RAW/g$i -o pmc --output-format csv -- python $R/bench.py --steps $STEPS --warmup $WARM --no-cpu-baseline --no-extras --no-steady > $RAW/g$i.json 2> $RAW/g$i.err
done
cd $R
python tools/summarize_valu.py $RAW $((STEPS + WARM)) $OUT
