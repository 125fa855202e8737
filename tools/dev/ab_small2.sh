# dev: two builds of the library on small frames (where sparse groups take the workgroup-per-keypoint forms), whole call, interleaved
cp sift_pyocl_amd/libsiftmi.so /tmp/libsiftmi_keep.so
for rep in 1 2 3; do for tag in "$@"; do
  cp sift_pyocl_amd/libsiftmi_$tag.so sift_pyocl_amd/libsiftmi.so
  echo "== $tag (rep $rep): 512 white: $(python tools/dev/ab_opts.py base=1 rounds=8 size=512 octaves=0 2>&1 | grep median | cut -c50-70) 512 smooth: $(python tools/dev/ab_opts.py base=1 rounds=8 size=512 octaves=0 kind=smooth 2>&1 | grep median | cut -c50-70) 1024 white: $(python tools/dev/ab_opts.py base=1 rounds=8 size=1024 octaves=0 2>&1 | grep median | cut -c50-70) 1024 smooth: $(python tools/dev/ab_opts.py base=1 rounds=8 size=1024 octaves=0 kind=smooth 2>&1 | grep median | cut -c50-70) 2048 white: $(python tools/dev/ab_opts.py base=1 rounds=8 size=2048 octaves=0 2>&1 | grep median | cut -c50-70)"
done; done
cp /tmp/libsiftmi_keep.so sift_pyocl_amd/libsiftmi.so
