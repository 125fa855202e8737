"""dev: where the host thread spends a BatchPlan call (64 x 2048^2): python tools/dev/batch_host.py [lanes]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import sift_pyocl_amd as sp
size = 2048
imgs = [torch.from_numpy(np.random.default_rng(i).random((size, size), dtype=np.float32)).cuda() for i in range(64)]
for lanes in [int(v) for v in sys.argv[1:]] or [8]:
    bp = sp.BatchPlan(shape=(size, size), dtype=np.float32, lanes=lanes)
    for _ in range(2): bp.keypoints_batch_device(imgs)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): bp.keypoints_batch_device(imgs)
    torch.cuda.synchronize()
    print("lanes %d: %.2f ms per batch of 64" % (lanes, 1e3 * (time.perf_counter() - t0) / 3), file=sys.stderr, flush=True)
    bp.set_option("host_timing", 1)
    bp.keypoints_batch_device(imgs)
    del bp
