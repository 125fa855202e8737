#!/usr/bin/env python
"""Where does the host time of one keypoints() call go?  (dev tool)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sift_pyocl_amd as sp
img = np.random.default_rng(0).random((4096, 4096), dtype=np.float32)
t = torch.from_numpy(img).cuda(); torch.cuda.synchronize()
plan = sp.SiftPlan(shape=img.shape, dtype=np.float32, profile="light", octave_max=3)
for _ in range(3): plan.keypoints(t)
a = []; b = []
for _ in range(10):
    t0 = time.perf_counter(); k = plan.keypoints(t); t1 = time.perf_counter(); kt = plan.kernel_times(); t2 = time.perf_counter()
    a.append(t1 - t0); b.append(t2 - t1)
print("keypoints() %.0f us  kernel_times() %.0f us  kernel span %.0f us" % (1e6 * np.median(a), 1e6 * np.median(b), 1e3 * kt["total_ms"]))
