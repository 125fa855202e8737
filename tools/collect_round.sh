#!/bin/bash
# Collect the rocprofv3 evidence of a round on the GPU box (from the repo root):  bash tools/collect_round.sh r03
# -> gpurun_out/round_<tag>/ ; copy what should be judged into profiles/<tag>/.
TAG=${1:-r06}
R=$(pwd)
OUT=$R/gpurun_out/round_$TAG
rm -rf $OUT; mkdir -p $OUT
bash tools/collect_profiles.sh $TAG > $OUT/collect.log 2>&1
cp gpurun_out/prof_$TAG/summary.txt $OUT/rocprofv3_summary.txt
cp gpurun_out/prof_$TAG/kt/*kernel_stats.csv $OUT/rocprofv3_kernel_stats.csv 2>/dev/null
cp gpurun_out/prof_$TAG/bench_under_rocprof.json $OUT/ 2>/dev/null
cp gpurun_out/prof_$TAG/blur_traffic.json $OUT/ 2>/dev/null
python tools/stage_profile.py 4096 white 3 > $OUT/stage_white4096.txt 2>&1
python tools/stage_profile.py 4096 smooth 0 > $OUT/stage_smooth4096.txt 2>&1
bash tools/dev/trace_gaps.sh > $OUT/timeline_white4096.txt 2>&1
for k in descriptor_kernel orientation_kernel; do bash tools/dev/pmc_split.sh $k > $OUT/pmc_$k.txt 2>&1; done   # (group 0, group 1) launches apart
bash tools/dev/pmc_kernel.sh extrema_kernel > $OUT/pmc_extrema_kernel.txt 2>&1
bash tools/dev/pmc_blur.sh > $OUT/pmc_blur_team_kernel.txt 2>&1                 # the dominant kernel: wait / issue / LDS counters per template instance
[ -x tools/ubench/valu2_bench ] && ./tools/ubench/valu2_bench > $OUT/valu2.txt 2>&1   # v_sad / f64 / dot4 issue rates (tools/ubench/valu2.hip)
[ -x tools/ubench/blur_var_bench ] && ./tools/ubench/blur_var_bench 4096 4096 384 > $OUT/blur_timeline.txt 2>&1   # per-wave timeline of the blur launches, priority feedback off / on
python tools/bench_match.py > $OUT/match_100k.txt 2>&1
python tools/dev/quick_smooth.py > $OUT/configs.txt 2>&1
python tools/dev/small_frames.py tail 1 0 sizes=256,512,1024,2048 > $OUT/small_frames.txt 2>&1
python tools/dev/small_frames.py tail 1 sizes=512,1024,2048 kind=smooth >> $OUT/small_frames.txt 2>&1
bash tools/dev/trace_small.sh 512 > $OUT/timeline_white512.txt 2>&1
bash tools/valu_frame.sh $OUT > $OUT/valu_frame.log 2>&1                       # VALU instructions per frame by kernel family and op class
./tools/ubench/valu_rate_bench > $OUT/valu_issue_rate.txt 2>&1                 # issue interval per op class (tools/ubench/valu_rate.hip)
python tools/bench_align.py > $OUT/bench_align.json 2>> $OUT/bench.err
# one kernel-stats summary per bench leg (the driver's line carries all of them; their kernels must not blend)
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $OUT/kt_c3 -o kt --output-format csv -- python $R/bench.py --size 16384 --octaves 0 --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-steady > $OUT/bench_c3_16384.json 2>> $OUT/bench.err )
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $OUT/kt_c4 -o kt --output-format csv -- python $R/bench.py --config c4 --steps 3 --warmup 1 > $OUT/bench_c4.json 2>> $OUT/bench.err )
for leg in c3 c4; do f=$(ls $OUT/kt_$leg/*/*kernel_stats.csv $OUT/kt_$leg/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $OUT/rocprofv3_kernel_stats_$leg.csv; rm -rf $OUT/kt_$leg; done
# HBM traffic of the 16384^2 blur launches (bench.py's c3 leg replays it)
bash tools/collect_c3_traffic.sh $TAG > $OUT/collect_c3.log 2>&1
cp gpurun_out/prof_c3_$TAG/blur_traffic.json $OUT/blur_traffic_c3.json 2>/dev/null
cp gpurun_out/prof_c3_$TAG/summary.txt $OUT/rocprofv3_summary_c3.txt 2>/dev/null
python bench.py > $OUT/bench.json 2> $OUT/bench.err
ls -la $OUT
# keep what travels back small (gpurun merges at most 64 MiB): the raw CSVs stay on the box
rm -rf gpurun_out/prof_$TAG gpurun_out/pmc_* gpurun_out/trace_gaps
