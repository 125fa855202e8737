#!/bin/bash
# dev: compare builds of the library (sift_pyocl_amd/libsiftmi_<tag>.so) on the same box, alternating processes, headline frame
#   bash tools/dev/ab_libs3.sh prev new
cp sift_pyocl_amd/libsiftmi.so /tmp/libsiftmi_keep.so
for rep in 1 2 3; do
  for tag in "$@"; do
    cp sift_pyocl_amd/libsiftmi_$tag.so sift_pyocl_amd/libsiftmi.so
    echo "== $tag (rep $rep): $(python tools/dev/ab_opts.py base=1 rounds=8 2>&1 | grep median)"
  done
done
cp /tmp/libsiftmi_keep.so sift_pyocl_amd/libsiftmi.so
