"""dev: BatchPlan time (64 x 2048^2, 8 lanes) for values of one option: python tools/dev/batch_opt.py march 1 0 [lanes=8] [size=2048]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import sift_pyocl_amd as sp
name = sys.argv[1]
vals = [int(v) for v in sys.argv[2:] if v.lstrip("-").isdigit()]
kw = dict(a.split("=") for a in sys.argv[2:] if "=" in a)
size = int(kw.get("size", 2048)); lanes = int(kw.get("lanes", 8)); n = int(kw.get("n", 64)); octaves = int(kw.get("octaves", 0))
imgs = [torch.from_numpy(np.random.default_rng(i).random((size, size), dtype=np.float32)).cuda() for i in range(n)]
for v in vals:
    bp = sp.BatchPlan(shape=(size, size), dtype=np.float32, lanes=lanes, octave_max=octaves or None)
    bp.set_option(name, v)
    for _ in range(2): bp.keypoints_batch_device(imgs)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(4): r = bp.keypoints_batch_device(imgs)
    torch.cuda.synchronize()
    print("%s=%-6d lanes %d: %.2f ms per batch of %d" % (name, v, lanes, 1e3 * (time.perf_counter() - t0) / 4, n), flush=True)
    del bp
