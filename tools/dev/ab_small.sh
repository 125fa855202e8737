#!/bin/bash
# dev: compare builds of the library (sift_pyocl_amd/libsiftmi_<tag>.so) on small frames:  bash tools/dev/ab_small.sh prev new
cp sift_pyocl_amd/libsiftmi.so /tmp/libsiftmi_keep.so
for rep in 1 2; do
  for tag in "$@"; do
    cp sift_pyocl_amd/libsiftmi_$tag.so sift_pyocl_amd/libsiftmi.so
    echo "== $tag (rep $rep)"; python tools/dev/small_frames.py tail 1 sizes=256,512,1024,2048 2>&1 | grep "\^2"
  done
done
cp /tmp/libsiftmi_keep.so sift_pyocl_amd/libsiftmi.so
