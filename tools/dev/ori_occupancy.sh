# dev: orientation launch of the headline frame's octave 0 alone (no overlap) against its workgroup count
for b in 304 608 1024 2048 4096; do
  echo "== ori_small_blocks=$b: $(python tools/stage_profile.py 4096 white 3 float32 overlap=0 ori_small_blocks=$b 2>&1 | grep -E 'orientation_assignment group 0')"
done
