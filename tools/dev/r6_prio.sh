#!/bin/bash
# round 6: the marching blur's priority feedback (option march_prio) -- stage parity, then interleaved A/B of whole calls
mkdir -p gpurun_out/r06
O=gpurun_out/r06/prio_ab.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "blur" > gpurun_out/r06/prio_tests.log 2>&1; echo "blur stage tests rc=$?" > $O
tail -2 gpurun_out/r06/prio_tests.log >> $O
for cfg in "size=4096 octaves=3" "size=4096 octaves=0" "size=2048 octaves=0" "size=2048 octaves=0 kind=smooth" "size=4096 octaves=0 kind=smooth" "size=1536 octaves=0"; do
  echo "== $cfg" >> $O
  python tools/dev/ab_flag.py opt=march_prio $cfg rounds=12 >> $O 2>&1
done
echo "== 16384 all octaves" >> $O
python tools/dev/ab_flag.py opt=march_prio size=16384 octaves=0 rounds=4 inner=3 >> $O 2>&1
cat $O
