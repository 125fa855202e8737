#!/bin/bash
# dev: the C4 batch (64 x 2048^2) and the headline frame against the number of hardware queues the HIP runtime maps streams to
# (GPU_MAX_HW_QUEUES, default 4; read by the runtime when it initialises) and the lanes' stream layout (overlap 0: one stream per lane)
for q in 4 8 16 24; do
  echo "== GPU_MAX_HW_QUEUES=$q"
  GPU_MAX_HW_QUEUES=$q python tools/dev/batch_opt.py overlap 0 1 0 1 lanes=16 2>&1 | grep -v amdgpu.ids
  GPU_MAX_HW_QUEUES=$q python tools/dev/batch_opt.py overlap 0 1 lanes=8 2>&1 | grep -v amdgpu.ids
  GPU_MAX_HW_QUEUES=$q python tools/dev/batch_opt.py overlap 1 lanes=2 size=4096 n=16 octaves=3 2>&1 | grep -v amdgpu.ids
  GPU_MAX_HW_QUEUES=$q python tools/dev/ab_opts.py base=1 rounds=8 2>&1 | grep median
  GPU_MAX_HW_QUEUES=$q python tools/dev/ab_opts.py base=1 size=512 octaves=0 rounds=8 2>&1 | grep median
done
