#!/bin/bash
# dev helper: run the stage profiler against the ablation build (libsiftmi_ablate.so)
cp sift_pyocl_amd/libsiftmi.so /tmp/libsiftmi_keep.so
cp sift_pyocl_amd/libsiftmi_ablate.so sift_pyocl_amd/libsiftmi.so
for a in 0 1 2 3; do echo "== ablate $a"; SIFTMI_ABLATE=$a python tools/stage_profile.py 4096 ${1:-white} 3 2>&1 | grep -E "descriptors  |TOTAL"; done
cp /tmp/libsiftmi_keep.so sift_pyocl_amd/libsiftmi.so
