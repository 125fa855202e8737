#!/bin/bash
# HBM traffic of the full-resolution blur launches at 16384^2 (BASELINE configs[2]: planes of 1 GiB, nothing fits the
# Infinity Cache): kernel trace + the two PMC passes of `bench.py --size 16384 --octaves 0`, summarised like the headline's.
# Usage (GPU box, repo root):  bash tools/collect_c3_traffic.sh r06   ->  gpurun_out/prof_c3_<tag>/{summary.txt, blur_traffic.json}
TAG=${1:-r06}
R=$(pwd)
OUT=$R/gpurun_out/prof_c3_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --size 16384 --octaves 0 --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-steady"      # 3 images
rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt --output-format csv -- $CMD > $OUT/bench_under_rocprof.json 2> $OUT/kt.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/fetch -o pmc --output-format csv -- $CMD > /dev/null 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/write -o pmc --output-format csv -- $CMD > /dev/null 2> $OUT/write.err
cd $R
python tools/summarize_prof.py $OUT $((16384 * 16384)) > $OUT/summary.txt 2>&1
grep -n "full-resolution\|blur family" $OUT/summary.txt
rm -rf $OUT/kt $OUT/fetch $OUT/write
