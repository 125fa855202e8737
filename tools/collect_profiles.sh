#!/bin/bash
# Collect the rocprofv3 evidence for bench.py: kernel trace + stats of the benchmark command itself,
# then FETCH_SIZE and WRITE_SIZE in two separate --pmc passes (kernel-trace only, as the pool requires).
# Usage (on the GPU box, from the repo root):  bash tools/collect_profiles.sh <tag>
TAG=${1:-r01}
R=$(pwd)
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras --no-steady"      # the headline workload only: 12 images
rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt --output-format csv -- $CMD > $OUT/bench_under_rocprof.json 2> $OUT/kt.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/fetch -o pmc --output-format csv -- $CMD > /dev/null 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/write -o pmc --output-format csv -- $CMD > /dev/null 2> $OUT/write.err
cd $R
python tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
