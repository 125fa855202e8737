"""dev: wall time of a keypoints() call on the headline frame against the hipEvent span of its kernels, and where the host's
share goes (Python wrapper vs the C call)"""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import sift_pyocl_amd as sp
from sift_pyocl_amd import _lib
size = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
img = np.random.default_rng(0).random((size, size), dtype=np.float32)
t = torch.from_numpy(img).cuda()
for prof in ("light", False):
    plan = sp.SiftPlan(shape=img.shape, dtype=np.float32, octave_max=3, profile=prof)
    for _ in range(60): plan.keypoints(t)
    if prof: plan.profile_totals(reset=True)
    n = 200
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): k = plan.keypoints(t)
    wall = (time.perf_counter() - t0) / n
    msg = "profile=%r: wall %.1f us per call" % (prof, 1e6 * wall)
    if prof:
        tt = plan.profile_totals()
        msg += ", kernels first->last %.1f us, host share %.1f us" % (1e3 * tt["total_ms"] / n, 1e6 * wall - 1e3 * tt["total_ms"] / n)
    print(msg)
    # the bare C call with a preallocated pinned array
    L = _lib.lib()
    out = _lib.pinned_empty(20000, plan.dtype_kp)
    nn, ovf = C.c_int64(0), C.c_int32(0)
    ptr = t.data_ptr()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): L.siftmi_plan_keypoints(plan._handle, ptr, 0, 1, out.ctypes.data, 2, 20000, C.byref(nn), C.byref(ovf))
    print("   bare C call: %.1f us per call" % (1e6 * (time.perf_counter() - t0) / n))
