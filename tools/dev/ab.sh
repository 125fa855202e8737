#!/bin/bash
# dev helper: time the descriptor / orientation stages with alternative builds of the library (sift_pyocl_amd/libsiftmi_<tag>.so)
cp sift_pyocl_amd/libsiftmi.so /tmp/libsiftmi_keep.so
for tag in "$@"; do
  if [ "$tag" != "base" ]; then cp sift_pyocl_amd/libsiftmi_$tag.so sift_pyocl_amd/libsiftmi.so; else cp /tmp/libsiftmi_keep.so sift_pyocl_amd/libsiftmi.so; fi
  for kind in white smooth; do
    echo "== $tag $kind"
    python tools/stage_profile.py 4096 $kind 3 float32 overlap=0 2>&1 | grep -E "descriptors group 0|orientation_assignment group 0|local_maxmin 0|TOTAL"
  done
done
cp /tmp/libsiftmi_keep.so sift_pyocl_amd/libsiftmi.so
