#!/bin/bash
mkdir -p gpurun_out/r06
O=gpurun_out/r06/prio_ab4.txt
: > $O
for cfg in "size=1024 octaves=0" "size=1536 octaves=0" "size=2048 octaves=0" "size=2048 octaves=0 kind=smooth" "size=3000 octaves=0" "size=4096 octaves=3" "size=4096 octaves=0" "size=4096 octaves=0 kind=smooth"; do
  echo "== march_prio 0 / 1 (default), $cfg" >> $O
  python tools/dev/ab_flag.py opt=march_prio vals=0,1 $cfg rounds=12 2>/dev/null >> $O
done
echo "== 16384" >> $O
python tools/dev/ab_flag.py opt=march_prio vals=0,1 size=16384 octaves=0 rounds=4 inner=3 2>/dev/null >> $O
cat $O
