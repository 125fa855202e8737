#!/bin/bash
mkdir -p gpurun_out/r06
O=gpurun_out/r06/later_prio2.txt
: > $O
echo "== later_prio x desc_small_blocks, headline (refine_spread 0)" >> $O
python tools/dev/ab_opts.py "base=1" "later_prio=1" "later_prio=2" "later_prio=1,desc_small_blocks=768" "later_prio=2,desc_small_blocks=768" "later_prio=1,desc_small_blocks=960" "later_prio=2,desc_small_blocks=960" "desc_small_blocks=768" "base=1" rounds=10 2>/dev/null >> $O
echo "== later_prio, 4096 all octaves" >> $O
python tools/dev/ab_opts.py "base=1" "later_prio=1" "later_prio=2" "later_prio=1,desc_small_blocks=960" "base=1" octaves=0 rounds=10 2>/dev/null >> $O
echo "== later_prio, 2048 all octaves" >> $O
python tools/dev/ab_opts.py "base=1" "later_prio=1" "later_prio=2" "base=1" size=2048 octaves=0 rounds=10 2>/dev/null >> $O
cat $O
