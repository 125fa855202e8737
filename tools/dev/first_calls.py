"""dev: wall time of each of the first calls of a fresh plan in a fresh process (which calls carry one-off costs?)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import sift_pyocl_amd as sp
size = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
prof = sys.argv[2] if len(sys.argv) > 2 else "light"
imgs = [torch.from_numpy(np.random.default_rng(i).random((size, size), dtype=np.float32)).cuda() for i in range(5)]
plan = sp.SiftPlan(shape=(size, size), dtype=np.float32, octave_max=3, profile=prof if prof != "none" else False)
torch.cuda.synchronize()
ts = []
for i in range(40):
    t1 = time.perf_counter(); k = plan.keypoints(imgs[i % 5]); ts.append(1e3 * (time.perf_counter() - t1))
print(" ".join("%.3f" % t for t in ts))
