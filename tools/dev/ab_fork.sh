#!/bin/bash
# dev: option A/B over a set of frames:  bash tools/dev/ab_fork.sh "fork=0" "base=1"
for cfg in "size=4096 octaves=4" "size=4096 octaves=5" "size=2048 octaves=3" "size=2048 octaves=0" "size=1024 octaves=0" "size=4096 octaves=0 kind=smooth" "size=2048 octaves=0 kind=smooth" "size=1024 octaves=0 kind=smooth"; do
  echo "== $cfg"; python tools/dev/ab_opts.py "$@" rounds=5 $cfg 2>&1 | grep median
done
