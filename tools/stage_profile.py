#!/usr/bin/env python
"""Per-stage device time (hipEvent) of one SiftPlan.keypoints() call: python tools/stage_profile.py [size] [white|smooth] [octaves] [dtype]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
import sift_pyocl_amd as sp
from util import smooth_noise, white_noise

size = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
kind = sys.argv[2] if len(sys.argv) > 2 else "white"
octaves = int(sys.argv[3]) if len(sys.argv) > 3 else 0
img = white_noise((size, size)) if kind == "white" else smooth_noise((size, size))
dtype = sys.argv[4] if len(sys.argv) > 4 else "float32"       # e.g. uint8, uint16 (typed frames)
if dtype != "float32":
    img = ((img - img.min()) / (img.max() - img.min()) * np.iinfo(dtype).max).astype(dtype)
t = torch.from_numpy(img).cuda()
plan = sp.SiftPlan(shape=img.shape, dtype=img.dtype, profile=True, octave_max=octaves or None)
for kv in sys.argv[5:]:                                        # name=value plan options (siftmi_plan_set_option)
    name, value = kv.split("=")
    plan.set_option(name, int(value))
for _ in range(3):
    kp = plan.keypoints(t)
print("image %s %dx%d octaves=%d -> %d keypoints" % (kind, size, size, plan.octave_max, len(kp)))
agg = {}
for label, ms in plan._profile_lines():
    key = label.split(" octave")[0].rstrip("0123456789 ") if label.startswith("Blur") else label.rstrip("0123456789 ")
    agg[key] = agg.get(key, 0.0) + ms
for label, ms in plan._profile_lines():
    if " 0" in label or "octave 0" in label or "normalize" in label or "max_min" in label:
        print("%45s %9.4f ms" % (label, ms))
print("-" * 60)
for k, v in sorted(agg.items(), key=lambda kv: -kv[1]):
    print("%45s %9.4f ms" % (k, v))
print("%45s %9.4f ms" % ("TOTAL kernels (first->last event)", plan.kernel_times()["total_ms"]))
