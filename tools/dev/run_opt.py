"""dev: n calls of one plan with the given options (for a run under rocprofv3):  python tools/dev/run_opt.py "xcd_map=0" [size=4096] [octaves=3] [n=6]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import sift_pyocl_amd as sp
sets = [a for a in sys.argv[1:] if not a.startswith(("size=", "octaves=", "n=", "kind="))]
kw = dict(a.split("=") for a in sys.argv[1:] if a.startswith(("size=", "octaves=", "n=", "kind=")))
size = int(kw.get("size", 4096)); octaves = int(kw.get("octaves", 3)); n = int(kw.get("n", 6))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from util import smooth_noise
img = smooth_noise((size, size)) if kw.get("kind") == "smooth" else np.random.default_rng(0).random((size, size), dtype=np.float32)
t = torch.from_numpy(img).cuda()
plan = sp.SiftPlan(shape=(size, size), dtype=np.float32, octave_max=octaves or None)
for s in sets:
    for kv in s.split(","):
        name, v = kv.split("="); plan.set_option(name, int(v))
for _ in range(n): kp = plan.keypoints(t)
print(len(kp))
