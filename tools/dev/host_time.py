import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch, time
import sift_pyocl_amd as sp
for size in (512, 2048):
    img = np.random.default_rng(0).random((size, size), dtype=np.float32)
    t = torch.from_numpy(img).cuda()
    plan = sp.SiftPlan(shape=img.shape, dtype=np.float32)
    for _ in range(5): k = plan.keypoints(t)
    plan.set_option("host_timing", 1)
    for _ in range(4):
        t0 = time.perf_counter(); k = plan.keypoints(t); print("call %.0f us" % (1e6 * (time.perf_counter() - t0)), file=sys.stderr)
