# dev: two builds of the library on the keypoint-rich frames (whole call, interleaved): bash tools/dev/ab_rich.sh old new
cp sift_pyocl_amd/libsiftmi.so /tmp/libsiftmi_keep.so
for rep in 1 2 3; do for tag in "$@"; do
  cp sift_pyocl_amd/libsiftmi_$tag.so sift_pyocl_amd/libsiftmi.so
  echo "== $tag (rep $rep): 4096 smooth all octaves: $(python tools/dev/ab_opts.py base=1 rounds=6 size=4096 octaves=0 kind=smooth 2>&1 | grep median | cut -c50-70)  2048 smooth: $(python tools/dev/ab_opts.py base=1 rounds=6 size=2048 octaves=0 kind=smooth 2>&1 | grep median | cut -c50-70)  maps=0 4096: $(python tools/dev/ab_opts.py maps=0 rounds=6 size=4096 octaves=0 kind=smooth 2>&1 | grep median | cut -c50-70)"
done; done
cp /tmp/libsiftmi_keep.so sift_pyocl_amd/libsiftmi.so
