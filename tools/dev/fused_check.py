"""dev: the fused orientation + description launch (option fused_kp) against the separate launches: same record SET
(records leave in no particular order), then interleaved timing."""
import os, sys, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import sift_pyocl_amd as sp
from util import smooth_noise

def canon(k):
    a = np.frombuffer(np.ascontiguousarray(k).tobytes(), dtype=np.uint8).reshape(len(k), -1)
    return a[np.lexsort(a.T[::-1])]

cases = [("white", 4096, 3), ("white", 2048, 0), ("smooth", 2048, 0), ("smooth", 1031, 0), ("white", 512, 0), ("smooth", 300, 0), ("white", 131, 0)]
bad = 0
for kind, size, octs in cases:
    shape = (size, size if size != 1031 else 1537)
    img = smooth_noise(shape) if kind == "smooth" else np.random.default_rng(size).random(shape, dtype=np.float32)
    t = torch.from_numpy(img).cuda()
    out = []
    for fused in (0, 1):
        plan = sp.SiftPlan(shape=img.shape, dtype=np.float32, octave_max=octs or None)
        plan.set_option("fused_kp", fused)
        for maps in ((2,) if size < 2048 else (0, 1)):
            plan.set_option("maps", maps)
            k = plan.keypoints(t); k = plan.keypoints(t)
            out.append(canon(k))
    same = all(o.shape == out[0].shape and np.array_equal(o, out[0]) for o in out)
    bad += not same
    print("%-6s %5d x %-5d octaves %d: %7d keypoints, %s" % (kind, shape[0], shape[1], octs, len(out[0]), "identical" if same else "DIFFERENT " + str([len(o) for o in out])), flush=True)
print("FAILED" if bad else "all identical")
