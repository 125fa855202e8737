"""dev: what the PCIe link gives for 64 MiB pinned -> device copies: one stream back to back, two streams alternating,
and the BatchPlan host pipeline at 2 / 3 / 4 lanes (ms per frame)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import sift_pyocl_amd as sp
n = 4096 * 4096
host = [torch.empty(n, dtype=torch.float32).pin_memory() for _ in range(4)]
dev = [torch.empty(n, dtype=torch.float32, device="cuda") for _ in range(4)]
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for mode in ("one stream", "two streams"):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(32):
        with torch.cuda.stream(s1 if (mode == "one stream" or i % 2 == 0) else s2):
            dev[i % 4].copy_(host[i % 4], non_blocking=True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("%-12s: %.3f ms per 64 MiB copy = %.1f GB/s" % (mode, 1e3 * dt / 32, 32 * n * 4 / dt / 1e9))
frames = [torch.from_numpy(np.random.default_rng(40 + i).random((4096, 4096), dtype=np.float32)).pin_memory().numpy() for i in range(16)]
import ctypes
L = sp._lib.lib()
L.hipMemcpyLike = None
raw = torch.empty(n, dtype=torch.float32, device="cuda")
plan1 = sp.SiftPlan(shape=(4096, 4096), dtype=np.float32, octave_max=3)
for _ in range(3): plan1.keypoints(frames[0])
t0 = time.perf_counter()
for i in range(8): plan1.keypoints(frames[i])
print("SiftPlan host frame, synchronous: %.3f ms per frame" % (1e3 * (time.perf_counter() - t0) / 8))
del plan1
for lanes, split in ((2, 0), (3, 0), (4, 0)):
    bp = sp.BatchPlan(shape=(4096, 4096), dtype=np.float32, octave_max=3, lanes=lanes)
    bp.keypoints_batch(frames)
    ts = []
    for _ in range(4):
        t0 = time.perf_counter(); bp.keypoints_batch(frames); ts.append(time.perf_counter() - t0)
    print("BatchPlan host frames, %d lanes, split %d: %.3f ms per frame (best %.3f)" % (lanes, split, 1e3 * sorted(ts)[1] / 16, 1e3 * min(ts) / 16))
    del bp
