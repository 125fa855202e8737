#!/bin/bash
# dev: builds sift_pyocl_amd/libsiftmi_<tag>.so against each other over a set of frames, alternating processes:
#   bash tools/dev/ab_libs_cfg.sh prev new -- "size=512 octaves=0" "size=2048 octaves=0"
tags=(); while [ "$1" != "--" ] && [ -n "$1" ]; do tags+=("$1"); shift; done; shift
cp sift_pyocl_amd/libsiftmi.so /tmp/libsiftmi_keep.so
for cfg in "$@"; do
  echo "== $cfg"
  for rep in $(seq 1 ${REPS:-2}); do for tag in "${tags[@]}"; do
    cp sift_pyocl_amd/libsiftmi_$tag.so sift_pyocl_amd/libsiftmi.so
    echo "   $tag: $(python tools/dev/ab_opts.py base=1 rounds=${ROUNDS:-6} $cfg 2>&1 | grep median)"
  done; done
done
cp /tmp/libsiftmi_keep.so sift_pyocl_amd/libsiftmi.so
