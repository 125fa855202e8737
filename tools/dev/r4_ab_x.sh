# dev: like r4_ab.sh, with the detection stage and small frames in the read-out
R=$(pwd); cp sift_pyocl_amd/libsiftmi.so /tmp/libsiftmi_keep.so
last="${@: -1}"
cp sift_pyocl_amd/libsiftmi_$last.so sift_pyocl_amd/libsiftmi.so
timeout 900 python -m pytest tests/test_gpu_params.py tests/test_gpu_fullsize.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -1
for rep in 1 2 3; do
  for tag in "$@"; do
    cp sift_pyocl_amd/libsiftmi_$tag.so sift_pyocl_amd/libsiftmi.so
    echo "== $tag (rep $rep): 4096/3: $(python tools/dev/ab_opts.py base=1 rounds=8 2>&1 | grep median | cut -c50-70) 2048: $(python tools/dev/ab_opts.py base=1 rounds=8 size=2048 octaves=0 2>&1 | grep median | cut -c50-70) 512: $(python tools/dev/ab_opts.py base=1 rounds=8 size=512 octaves=0 2>&1 | grep median | cut -c50-70)"
  done
done
for tag in "$@"; do
  cp sift_pyocl_amd/libsiftmi_$tag.so sift_pyocl_amd/libsiftmi.so
  echo "== $tag stages, white 4096 / 3 octaves"
  python tools/stage_profile.py 4096 white 3 float32 overlap=0 2>&1 | grep -E "local_maxmin 0|interp_keypoint.* 0|TOTAL"
done
cp /tmp/libsiftmi_keep.so sift_pyocl_amd/libsiftmi.so
