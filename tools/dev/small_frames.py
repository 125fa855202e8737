"""dev: wall time per keypoints() call on small frames (all octaves) for values of one plan option:
   python tools/dev/small_frames.py tile 1 0 [kind=white|smooth] [sizes=256,512,1024,2048]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import sift_pyocl_amd as sp
from util import smooth_noise
name = sys.argv[1]
vals = [int(v) for v in sys.argv[2:] if v.lstrip("-").isdigit()]
kw = dict(a.split("=") for a in sys.argv[2:] if "=" in a)
kind = kw.get("kind", "white")
sizes = [int(v) for v in kw.get("sizes", "256,512,1024,2048").split(",")]
for size in sizes:
    img = smooth_noise((size, size)) if kind == "smooth" else np.random.default_rng(0).random((size, size), dtype=np.float32)
    t = torch.from_numpy(img).cuda()
    ref = None
    for v in vals:
        plan = sp.SiftPlan(shape=img.shape, dtype=np.float32)
        plan.set_option(name, v)
        for _ in range(5): k = plan.keypoints(t)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): k = plan.keypoints(t)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
        rows = np.ascontiguousarray(np.asarray(k)).view(np.uint8).reshape(len(k), -1)
        raw = rows[np.lexsort(rows.T[::-1])].tobytes() if len(k) else b''
        same = "" if ref is None else ("same" if raw == ref else "DIFFERENT")
        if ref is None: ref = raw
        print("%5d^2 %s=%-5d %8.4f ms  %6d kp %s" % (size, name, v, 1e3 * dt, len(k), same), flush=True)
        del plan
