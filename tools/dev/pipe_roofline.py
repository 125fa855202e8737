import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import sift_pyocl_amd as sp
S = 4096
dev = [torch.from_numpy(np.random.default_rng(i).random((S, S), dtype=np.float32)).cuda() for i in range(8)]
frames = [dev[i % 8] for i in range(20)]
for lanes in (1, 2, 3):
    bp = sp.BatchPlan(shape=(S, S), dtype=np.float32, octave_max=3, lanes=lanes, profile="light")
    bp.keypoints_batch(frames)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    bp.keypoints_batch(frames)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    bt = bp.blur_times()
    gbs = 8.0 * bt["blur0_pixels"] / 1e9 / (bt["blur0_ms"] / 1e3)
    print("lanes %d: %.3f ms/frame %.0f Mpix/s | blur brackets: %d launches %.1f us avg -> %.0f GB/s = %.3f of 8 TB/s" % (
        lanes, 1e3 * dt / len(frames), len(frames) * S * S / 1e6 / dt, bt["blur0_launches"], 1e3 * bt["blur0_ms"] / bt["blur0_launches"], gbs, gbs / 8000))
    del bp
