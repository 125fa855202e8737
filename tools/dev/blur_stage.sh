#!/bin/bash
# dev: the six full-resolution blur launches of the headline frame, each alone between events (stage profile), for builds
# sift_pyocl_amd/libsiftmi_<tag>.so:  bash tools/dev/blur_stage.sh base w4
cp sift_pyocl_amd/libsiftmi.so /tmp/libsiftmi_keep.so
for rep in 1 2; do for tag in "$@"; do
  cp sift_pyocl_amd/libsiftmi_$tag.so sift_pyocl_amd/libsiftmi.so
  echo "== $tag (rep $rep): $(python tools/stage_profile.py 4096 white 3 float32 overlap=0 2>&1 | grep -E 'initial blur|Blur octave 0' | awk '{printf "%s ", $(NF-1)*1000}')"
done; done
cp /tmp/libsiftmi_keep.so sift_pyocl_amd/libsiftmi.so
