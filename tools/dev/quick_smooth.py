import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np, torch
import sift_pyocl_amd as sp
from util import smooth_noise
for size, kind, octs in ((4096, "smooth", 0), (2048, "smooth", 0), (4096, "white", 0), (4096, "white", 3), (2048, "white", 0)):
    img = smooth_noise((size, size)) if kind == "smooth" else np.random.default_rng(0).random((size, size), dtype=np.float32)
    t = torch.from_numpy(img).cuda()
    plan = sp.SiftPlan(shape=img.shape, dtype=np.float32, octave_max=octs or None)
    for _ in range(3): k = plan.keypoints(t)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): k = plan.keypoints(t)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print("%5d %-6s oct %d: %8.3f ms  %6d kp" % (size, kind, plan.octave_max, 1e3 * dt, len(k)))
    del plan
