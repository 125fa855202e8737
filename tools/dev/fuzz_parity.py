"""dev: randomized HIP-vs-oracle parity over odd shapes / dtypes (exercises the team blur kernel's edges)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import sift_pyocl_amd as sp
from oracle import pyoracle
from util import assert_same_keypoints, smooth_noise, white_noise
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
t0 = time.time()
for it in range(n):
    H = int(rng.integers(512, 1700)); W = int(rng.integers(1024, 2300))
    if it % 5 == 0: H, W = W, H                       # tall frames: W may drop below 1024 -> tile kernel
    kind = it % 3
    img = white_noise((H, W), seed=it) if kind == 0 else smooth_noise((H, W), seed=it, sigma=1.0 + (it % 4))
    dt = [np.float32, np.uint8, np.uint16, np.float32][it % 4]
    if dt != np.float32:
        img = ((img - img.min()) / (img.max() - img.min()) * np.iinfo(dt).max).astype(dt)
    want = pyoracle.keypoints(img.astype(np.float32))
    plan = sp.SiftPlan(template=img)
    got = plan.keypoints(img)
    assert_same_keypoints(got, want, "fuzz %d %dx%d %s" % (it, H, W, np.dtype(dt).name))
    plan.set_option("maps", 1)                        # full gradient maps, whatever the density
    assert_same_keypoints(plan.keypoints(img), want, "fuzz %d %dx%d %s, gradient maps" % (it, H, W, np.dtype(dt).name))
    if it % 7 == 0:
        bp = sp.BatchPlan(template=img, lanes=2)
        for g in bp.keypoints_batch([img, img]):
            assert_same_keypoints(g, want, "fuzz batch %d" % it)
    print("ok %2d  %4dx%4d %-7s %6d kp" % (it, H, W, np.dtype(dt).name, len(got)), flush=True)
print("all %d cases bit-identical (%.0f s)" % (n, time.time() - t0))
