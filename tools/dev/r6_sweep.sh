#!/bin/bash
# round 6: the schedule options re-swept on the headline frame with the faster blur launches (interleaved A/B, one process per line group)
mkdir -p gpurun_out/r06
O=gpurun_out/r06/sweep.txt
: > $O
python tools/dev/ab_opts.py "base=1" "early_chain=1" "fork=1" "split=1" "fused_refine=2" "base=1" rounds=10 >> $O 2>/dev/null
python tools/dev/ab_opts.py "base=1" "desc_small_blocks=448" "desc_small_blocks=512" "desc_small_blocks=640" "desc_small_blocks=704" "desc_small_blocks=768" "desc_small_blocks=960" rounds=10 >> $O 2>/dev/null
python tools/dev/ab_opts.py "base=1" "ori_small_blocks=448" "ori_small_blocks=768" "ori_small_blocks=1024" "desc_team=1024" "desc_dynamic=0" "ext_strips=1000" "ext_strips=4000" rounds=10 >> $O 2>/dev/null
python tools/dev/ab_opts.py "base=1" "march_wgs=640" "march_wgs=896" "march_wgs=1024" "mm_blocks=512" "maps=0" rounds=10 >> $O 2>/dev/null
cat $O
