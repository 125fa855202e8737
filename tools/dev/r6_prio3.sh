#!/bin/bash
mkdir -p gpurun_out/r06
O=gpurun_out/r06/prio_ab3.txt
: > $O
for cfg in "size=4096 octaves=3" "size=4096 octaves=0" "size=4096 octaves=0 kind=smooth" "size=3000 octaves=0"; do
  echo "== march_prio 0 / 1 (N>=15, >=704 wgs) / 3 (>=704 wgs) / 2 (always), $cfg" >> $O
  python tools/dev/ab_flag.py opt=march_prio vals=0,1,3,2 $cfg rounds=12 2>/dev/null >> $O
done
echo "== 16384" >> $O
python tools/dev/ab_flag.py opt=march_prio vals=0,1,3,2 size=16384 octaves=0 rounds=4 inner=3 2>/dev/null >> $O
cat $O
