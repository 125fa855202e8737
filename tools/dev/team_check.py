import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import sift_pyocl_amd as sp
from util import smooth_noise, assert_same_keypoints
for shape in [(300, 517), (512, 512), (1030, 770), (2048, 2048)]:
    img = smooth_noise(shape, seed=5, sigma=2.0)
    a = sp.SiftPlan(template=img); a.set_option("desc_team", 0)
    b = sp.SiftPlan(template=img); b.set_option("desc_team", 1 << 30)
    ka, kb = a.keypoints(img), b.keypoints(img)
    assert_same_keypoints(ka, kb, "team vs wave %r" % (shape,))
    print(shape, len(ka), "same", flush=True)
