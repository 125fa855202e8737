#!/bin/bash
# dev: two builds of the library (sift_pyocl_amd/libsiftmi_<tag>.so) on the same box, alternating processes, on chosen frames:
#   bash tools/dev/ab_libs_cfg2.sh "size=4096 octaves=0|size=4096 octaves=3" head new
CFGS=$1; shift
cp sift_pyocl_amd/libsiftmi.so /tmp/libsiftmi_keep.so
for rep in 1 2 3; do
  for tag in "$@"; do
    cp sift_pyocl_amd/libsiftmi_$tag.so sift_pyocl_amd/libsiftmi.so
    echo "== $tag (rep $rep)"
    IFS='|' read -ra CS <<< "$CFGS"
    for c in "${CS[@]}"; do python tools/dev/ab_opts.py base=1 $c rounds=8 2>&1 | grep median | sed "s/^/  $c  /"; done
  done
done
cp /tmp/libsiftmi_keep.so sift_pyocl_amd/libsiftmi.so
