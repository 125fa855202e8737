#!/bin/bash
# dev: kernel time of the team blur (11 and 27 taps, 4096 columns, several heights) for ablation builds libsiftmi_b<bits>.so
# bits: 1 no H pass, 2 V pass with one product, 4 no global loads, 8 no global stores
R=$(pwd)
cp sift_pyocl_amd/libsiftmi.so /tmp/libsiftmi_keep.so
cd /tmp && export TMPDIR=/tmp
for tag in "$@"; do
  if [ "$tag" != "base" ]; then cp $R/sift_pyocl_amd/libsiftmi_$tag.so $R/sift_pyocl_amd/libsiftmi.so; else cp /tmp/libsiftmi_keep.so $R/sift_pyocl_amd/libsiftmi.so; fi
  rm -rf /tmp/bs; rocprofv3 --kernel-trace -d /tmp/bs -o kt --output-format csv -- python $R/tools/dev/blur_scaling.py > /dev/null 2>&1
  echo "== $tag"; python $R/tools/dev/kt_print.py /tmp/bs blur_team | awk '{print $3, $(NF-3), $(NF-1)}' | sed 's/siftk::blur_team_kernel//' | awk 'NR%2==0' | tr '\n' ';'; echo
done
cp /tmp/libsiftmi_keep.so $R/sift_pyocl_amd/libsiftmi.so
