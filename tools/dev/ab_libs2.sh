#!/bin/bash
# dev: two builds (sift_pyocl_amd/libsiftmi_<tag>.so), whole calls on the headline frame and the dense frame, alternating
cp sift_pyocl_amd/libsiftmi.so /tmp/libsiftmi_keep.so
for rep in 1 2 3; do
  for tag in "$@"; do
    cp sift_pyocl_amd/libsiftmi_$tag.so sift_pyocl_amd/libsiftmi.so
    echo "== $tag (rep $rep)"
    python tools/dev/ab_opts.py base=1 rounds=8 2>&1 | grep median | sed 's/^/  4096 white  /'
    python tools/dev/ab_opts.py base=1 kind=smooth rounds=3 2>&1 | grep median | sed 's/^/  4096 smooth /'
  done
done
cp /tmp/libsiftmi_keep.so sift_pyocl_amd/libsiftmi.so
