#!/bin/bash
mkdir -p gpurun_out/r06
O=gpurun_out/r06/prio_ab2.txt
: > $O
for cfg in "size=4096 octaves=3" "size=4096 octaves=0" "size=2048 octaves=0" "size=4096 octaves=0 kind=smooth"; do
  echo "== ext_prio, $cfg" >> $O
  python tools/dev/ab_flag.py opt=ext_prio $cfg rounds=12 2>/dev/null >> $O
done
for cfg in "size=2048 octaves=0" "size=1536 octaves=0" "size=4096 octaves=3"; do
  echo "== march_prio 0 / 1 (rule) / 2 (forced), $cfg" >> $O
  python tools/dev/ab_flag.py opt=march_prio vals=0,1,2 $cfg rounds=12 2>/dev/null >> $O
done
cat $O
