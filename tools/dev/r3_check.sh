#!/bin/bash
# dev (round 3): parity subset + bench + descriptor stage time on one box
R=$(pwd); OUT=$R/gpurun_out/r3chk; rm -rf $OUT; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_params.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu > $OUT/pytest.log 2>&1
tail -5 $OUT/pytest.log
python bench.py --no-cpu-baseline --no-extras > $OUT/bench.json 2> $OUT/bench.err
python tools/stage_profile.py 4096 white 3 float32 overlap=0 2>&1 | grep -E "descriptors group|orientation_assignment group|TOTAL|keypoints|local_maxmin 0|interp" > $OUT/stage_white.txt
python tools/stage_profile.py 4096 smooth 0 float32 overlap=0 2>&1 | grep -E "descriptors group|orientation_assignment group|TOTAL|keypoints" > $OUT/stage_smooth.txt
cat $OUT/stage_white.txt $OUT/stage_smooth.txt
python -c "import json;d=json.load(open('$OUT/bench.json'));print(d['value'],d['ms_per_step'],d['roofline']['frac'])"
