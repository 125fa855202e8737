"""dev: Python's share of a keypoints() call: the method against the bare C call into a reused pinned array"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import ctypes as C, numpy as np, torch
import sift_pyocl_amd as sp
from sift_pyocl_amd import _lib
size = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
octs = int(sys.argv[2]) if len(sys.argv) > 2 else 3
img = np.random.default_rng(0).random((size, size), dtype=np.float32)
t = torch.from_numpy(img).cuda(); torch.cuda.synchronize()
plan = sp.SiftPlan(shape=img.shape, dtype=np.float32, octave_max=octs or None)
L = _lib.lib()
for _ in range(30): k = plan.keypoints(t)
cap = int(1.5 * len(k)) + 256
out = _lib.pinned_empty(cap, plan.dtype_kp)
n = C.c_int64(); ovf = C.c_int32()
ptr = t.data_ptr(); optr = out.ctypes.data; h = plan._handle
for rep in range(3):
    N = 200
    t0 = time.perf_counter()
    for _ in range(N): k = plan.keypoints(t)
    t1 = time.perf_counter()
    for _ in range(N): L.siftmi_plan_keypoints(h, ptr, 0, 1, optr, 2, cap, C.byref(n), C.byref(ovf))
    t2 = time.perf_counter()
    print("%d^2: method %.1f us, bare C call %.1f us, difference %.1f us" % (size, 1e6 * (t1 - t0) / N, 1e6 * (t2 - t1) / N, 1e6 * ((t1 - t0) - (t2 - t1)) / N), flush=True)
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(200): k = plan.keypoints(t)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
