"""dev: wall time per keypoints() call on the headline frame for values of one plan option:
   python tools/dev/sweep_opt.py desc_pad 0 10000 20000 [size] [kind]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import sift_pyocl_amd as sp
from util import smooth_noise
name = sys.argv[1]
vals = [int(v) for v in sys.argv[2:] if v.lstrip("-").isdigit()]
rest = [v for v in sys.argv[2:] if not v.lstrip("-").isdigit()]
kind = rest[0] if rest else "white"
size = 4096
img = smooth_noise((size, size)) if kind == "smooth" else np.random.default_rng(0).random((size, size), dtype=np.float32)
t = torch.from_numpy(img).cuda()
for v in vals:
    plan = sp.SiftPlan(shape=img.shape, dtype=np.float32, octave_max=3)
    plan.set_option(name, v)
    for _ in range(5): k = plan.keypoints(t)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): k = plan.keypoints(t)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 30
    print("%s=%-7d %8.4f ms  %6d kp" % (name, v, 1e3 * dt, len(k)), flush=True)
    del plan
