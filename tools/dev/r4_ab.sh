#!/bin/bash
# dev (round 4): builds of the library (sift_pyocl_amd/libsiftmi_<tag>.so) against each other on ONE box, alternating processes:
# whole call on the headline frame + the keypoint stages alone (white 4096^2 / 3 octaves, smoothed 4096^2 / all octaves).
#   bash tools/dev/r4_ab.sh [-t] prev new ...     (-t: parity subset with the LAST tag first)
R=$(pwd); OUT=$R/gpurun_out/r4ab; mkdir -p $OUT
cp sift_pyocl_amd/libsiftmi.so /tmp/libsiftmi_keep.so
if [ "$1" = "-t" ]; then
  shift; last="${@: -1}"
  cp sift_pyocl_amd/libsiftmi_$last.so sift_pyocl_amd/libsiftmi.so
  timeout 900 python -m pytest tests/test_gpu_params.py tests/test_gpu_fullsize.py tests/test_gpu_parity.py -x -q -m gpu > $OUT/pytest_$last.log 2>&1
  echo "== parity subset ($last): $(tail -1 $OUT/pytest_$last.log)"
fi
for rep in 1 2 3; do
  for tag in "$@"; do
    cp sift_pyocl_amd/libsiftmi_$tag.so sift_pyocl_amd/libsiftmi.so
    echo "== $tag (rep $rep): $(python tools/dev/ab_opts.py base=1 rounds=8 2>&1 | grep median)"
  done
done
for tag in "$@"; do
  cp sift_pyocl_amd/libsiftmi_$tag.so sift_pyocl_amd/libsiftmi.so
  echo "== $tag stages, white 4096 / 3 octaves"
  python tools/stage_profile.py 4096 white 3 float32 overlap=0 2>&1 | grep -E "descriptors group|orientation_assignment group|local_maxmin 0|TOTAL"
  echo "== $tag stages, smooth 4096 / all octaves"
  python tools/stage_profile.py 4096 smooth 0 float32 overlap=0 2>&1 | grep -E "descriptors group 0|orientation_assignment group 0|TOTAL|keypoints"
done
cp /tmp/libsiftmi_keep.so sift_pyocl_amd/libsiftmi.so
