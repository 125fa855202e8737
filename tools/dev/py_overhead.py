"""dev: where the host time of a small-frame keypoints() call goes (cProfile over 1000 calls on a 256^2 frame)"""
import os, sys, time, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import sift_pyocl_amd as sp
size = int(sys.argv[1]) if len(sys.argv) > 1 else 256
img = np.random.default_rng(0).random((size, size), dtype=np.float32)
t = torch.from_numpy(img).cuda()
plan = sp.SiftPlan(shape=img.shape, dtype=np.float32)
for _ in range(20): plan.keypoints(t)
t0 = time.perf_counter()
for _ in range(1000): plan.keypoints(t)
print("plain loop: %.1f us per call" % (1e3 * (time.perf_counter() - t0)))
pr = cProfile.Profile(); pr.enable()
for _ in range(1000): plan.keypoints(t)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
