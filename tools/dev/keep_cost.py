"""dev: what the per-step parking of the records costs in bench.py's N > 1 line (one D->D copy out of the plan's list)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import ctypes as C, numpy as np, torch
import sift_pyocl_amd as sp
from sift_pyocl_amd import _lib
from bench import make_image
size = 4096
plan = sp.SiftPlan(shape=(size, size), dtype=np.float32, octave_max=3, profile="light")
imgs = [torch.from_numpy(make_image(i, size)).cuda() for i in range(5)]
arena = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
L = _lib.lib()
def run(mode, n=200):
    used = 0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        k = plan.keypoints(imgs[i % 5])
        if mode == 1:
            rv = plan.device_records(); m = rv.count
            arena[used:used + m * 144] = torch.as_tensor(rv, device="cuda")
            used = (used + m * 144) % (32 << 20)
        elif mode == 2:
            m = len(k)
            L.siftmi_plan_fetch(plan._handle, C.c_void_p(arena.data_ptr() + used), 1, 0, m)
            used = (used + m * 144) % (32 << 20)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n
for _ in range(60): plan.keypoints(imgs[0])
for rep in range(3):
    print("plain %.4f ms   torch slice copy %.4f ms   siftmi_plan_fetch to device %.4f ms" % (run(0), run(1), run(2)), flush=True)
