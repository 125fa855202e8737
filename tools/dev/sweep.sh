#!/bin/bash
# dev: sweep launch knobs of the per-keypoint kernels on the bench workload (ms per image, median of 3 runs)
run() { for i in 1 2 3; do env "$@" python bench.py --no-cpu-baseline --no-pipelined --steps 30 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])"; done | sort -n | sed -n 2p; }
echo "default            $(run X=1)"
for v in 12000 16000 24000; do echo "DESC_PAD $v     $(run SIFTMI_DESC_PAD=$v)"; done
for v in 1536 3072; do echo "DESC_BLOCKS $v   $(run SIFTMI_DESC_BLOCKS=$v)"; done
for v in 20000 40000; do echo "ORI_PAD $v      $(run SIFTMI_ORI_PAD=$v)"; done
for v in 512 2048; do echo "ORI_BLOCKS $v     $(run SIFTMI_ORI_BLOCKS=$v)"; done
