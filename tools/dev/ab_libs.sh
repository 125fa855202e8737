#!/bin/bash
# dev: compare two builds of the library (sift_pyocl_amd/libsiftmi_<tag>.so) on the same box, alternating processes:
#   bash tools/dev/ab_libs.sh prev new
cp sift_pyocl_amd/libsiftmi.so /tmp/libsiftmi_keep.so
for rep in 1 2 3; do
  for tag in "$@"; do
    cp sift_pyocl_amd/libsiftmi_$tag.so sift_pyocl_amd/libsiftmi.so
    echo "== $tag (rep $rep)"
    python tools/dev/ab_opts.py base=1 size=512 octaves=0 rounds=10 2>&1 | grep median | sed 's/^/  512   /'
    python tools/dev/ab_opts.py base=1 size=2048 octaves=0 rounds=8 2>&1 | grep median | sed 's/^/  2048  /'
    python tools/dev/ab_opts.py base=1 rounds=8 2>&1 | grep median | sed 's/^/  4096  /'
  done
done
cp /tmp/libsiftmi_keep.so sift_pyocl_amd/libsiftmi.so
