#!/bin/bash
run() { for i in 1 2 3; do env "$@" python bench.py --no-cpu-baseline --no-pipelined --steps 30 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])"; done | sort -n | sed -n 2p; }
echo "default            $(run X=1)"
echo "NO_PRIO            $(run SIFTMI_NO_PRIO=1)"
echo "NO_TEAM            $(run SIFTMI_NO_TEAM=1)"
echo "DESC_PAD 0         $(run SIFTMI_DESC_PAD=0)"
