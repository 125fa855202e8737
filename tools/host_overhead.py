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

# the two C calls of keypoints() separately: count (enqueue + wait), then fetch (D->H of the records)
import ctypes as C
from sift_pyocl_amd import _lib
L = _lib.lib()
n = C.c_int64(); ovf = C.c_int32()
tc = []; tf = []; ta = []
for _ in range(10):
    t0 = time.perf_counter()
    L.siftmi_plan_keypoints(plan._handle, t.data_ptr(), 0, 1, None, 0, 0, C.byref(n), C.byref(ovf))
    t1 = time.perf_counter()
    out = np.empty(n.value, dtype=plan.dtype_kp)
    t2 = time.perf_counter()
    L.siftmi_plan_fetch(plan._handle, out.ctypes.data, 0, 0, n.value)
    t3 = time.perf_counter()
    tc.append(t1 - t0); ta.append(t2 - t1); tf.append(t3 - t2)
print("count call %.0f us, np.empty %.0f us, fetch %d records (%.2f MB) %.0f us" % (1e6 * np.median(tc), 1e6 * np.median(ta), n.value, n.value * 144 / 1e6, 1e6 * np.median(tf)))
