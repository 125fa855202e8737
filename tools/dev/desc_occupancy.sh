for b in 144 288 432 576 768 960; do
  echo "== desc_small_blocks=$b: $(python tools/stage_profile.py 4096 white 3 float32 overlap=0 desc_small_blocks=$b 2>&1 | grep -E 'descriptors group 0')"
done
for b in 288 576 960; do
  echo "== dense desc blocks=$b: $(python tools/stage_profile.py 4096 smooth 0 float32 overlap=0 desc_dense_blocks=$b desc_blocks=$b 2>&1 | grep -E 'descriptors group 0')"
done
