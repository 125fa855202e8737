import sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import sift_pyocl_amd as sp
from oracle import pyoracle as o
from util import smooth_noise
for seed in (0, 1, 2, 3, 4):
    img = smooth_noise((512, 512), seed=seed)
    par = o.default_params(); par.pix_per_kp = 120
    want, ovf = o.keypoints(img, par, return_overflow=True)
    plan = sp.SiftPlan(template=img, PIX_PER_KP=120)
    got = plan.keypoints(img)
    print(seed, len(want), ovf, len(got), plan.overflow, plan.capacity(), flush=True)
imgs = [smooth_noise((512, 512), seed=s) for s in (0, 1, 2, 3, 4)]
for lanes in (1, 2):
    bp = sp.BatchPlan(template=imgs[0], PIX_PER_KP=120, lanes=lanes)
    out = bp.keypoints_batch(imgs)
    print("batch lanes", lanes, [len(x) for x in out], bp.overflow)
    out = bp.keypoints_batch(imgs)
    print("batch again", lanes, [len(x) for x in out], bp.overflow)
