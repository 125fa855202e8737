"""dev: per-step wall times of the bench loop (profile='light', 8 rotating frames, kernel_times() per step)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import sift_pyocl_amd as sp
size = 4096
imgs = [torch.from_numpy(np.random.default_rng(i).random((size, size), dtype=np.float32)).cuda() for i in range(8)]
plan = sp.SiftPlan(shape=(size, size), dtype=np.float32, octave_max=3, profile="light")
for i in range(5): plan.keypoints(imgs[i % 8])
torch.cuda.synchronize()
ts = []
t0 = time.perf_counter()
for i in range(40):
    t1 = time.perf_counter()
    k = plan.keypoints(imgs[i % 8]); plan.kernel_times()
    ts.append(1e3 * (time.perf_counter() - t1))
torch.cuda.synchronize()
print("total/40 = %.4f ms" % (1e3 * (time.perf_counter() - t0) / 40))
print(" ".join("%.3f" % t for t in ts))
