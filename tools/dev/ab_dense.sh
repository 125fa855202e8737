cp sift_pyocl_amd/libsiftmi.so /tmp/libsiftmi_keep.so
for rep in 1 2; do
  for tag in "$@"; do
    cp sift_pyocl_amd/libsiftmi_$tag.so sift_pyocl_amd/libsiftmi.so
    echo "== $tag (rep $rep): $(python tools/dev/ab_opts.py base=1 rounds=6 kind=smooth size=4096 octaves=0 2>&1 | grep median)"
  done
done
cp /tmp/libsiftmi_keep.so sift_pyocl_amd/libsiftmi.so
