#!/bin/bash
# dev helper: run the stage profiler against the ablation build (libsiftmi_ablate.so); args: image kind, ablate codes...
cp sift_pyocl_amd/libsiftmi.so /tmp/libsiftmi_keep.so
cp gpurun_in_libsiftmi_ablate.so sift_pyocl_amd/libsiftmi.so
kind=${1:-white}; shift
for a in "$@"; do echo "== ablate $a"; SIFTMI_ABLATE=$a python tools/stage_profile.py 4096 $kind 3 float32 overlap=0 2>&1 | grep -E "descriptors group 0|orientation_assignment group 0"; done
cp /tmp/libsiftmi_keep.so sift_pyocl_amd/libsiftmi.so
