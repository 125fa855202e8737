"""dev: the C4 batch (64 x 2048^2, 16 lanes) against the workgroup count of the marching blur (option march_wgs):
with 16 lanes in flight the chip is full anyway, so taller segments (less warm-up) may pay: python tools/dev/c4_wgs.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import sift_pyocl_amd as sp
from bench import make_image
size = 2048
frames = [torch.from_numpy(make_image(1000 + i, size)).cuda() for i in range(64)]
opts = [a.split("=") for a in sys.argv[1:]] or None
for wgs in (0, 640, 512, 384, 256, 192, 128):
    bp = sp.BatchPlan(shape=(size, size), dtype=np.float32)
    bp.set_option("march_wgs", wgs)
    for _ in range(2): bp.keypoints_batch_device(frames)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); counts, rec = bp.keypoints_batch_device(frames); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print("march_wgs %4d: %.2f ms per batch (min %.2f)  %d keypoints" % (wgs, 1e3 * sorted(ts)[2], 1e3 * min(ts), sum(counts)), flush=True)
    del bp
