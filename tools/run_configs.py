#!/usr/bin/env python
"""Measure the BASELINE.json configurations that fit one GPU (fills the table in BASELINE.md section 4).
python tools/run_configs.py [--skip-16k]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import sift_pyocl_amd as sp
from sift_pyocl_amd.batch import keypoints_batch
from util import smooth_noise

def bytes_alg(w, h, n_oct, n_kp):
    return w * h * (12.0 + 66.0 * sum(4.0 ** -o for o in range(n_oct))) + 144.0 * n_kp

def timed(plan, img, reps):
    for _ in range(2): k = plan.keypoints(img)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): k = plan.keypoints(img)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps, k

def run(size, kind, octaves, reps, device_input=True):
    rng = np.random.default_rng(0 if kind == "white" else 3)
    img = rng.random((size, size), dtype=np.float32) if kind == "white" else smooth_noise((size, size))
    plan = sp.SiftPlan(shape=img.shape, dtype=np.float32, octave_max=octaves or None, profile="light")
    src = torch.from_numpy(img).cuda() if device_input else img
    dt, k = timed(plan, src, reps)
    kt = plan.kernel_times()
    if kt["total_ms"] <= 0: kt["total_ms"] = 1e3 * dt      # light profile: only the blur bracket is timed; the wall time stands in
    ba = bytes_alg(size, size, plan.octave_max, len(k))
    print("| %5d^2 %-6s | oct %d | %s | %9.3f ms | %8.0f Mpix/s | %9.0f kp/s | %7d kp | kernels %.3f ms | pipeline %.0f GB/s (%.1f %% of 8 TB/s) | blur oct0 %.0f GB/s |" % (
        size, kind, plan.octave_max, "device" if device_input else "host  ", 1e3 * dt, size * size / 1e6 / dt, len(k) / dt, len(k),
        kt["total_ms"], ba / kt["total_ms"] / 1e6, 100 * ba / kt["total_ms"] / 1e6 / 8000, 8 * kt["blur0_pixels"] / max(kt["blur0_ms"], 1e-9) / 1e6), flush=True)
    del plan

print("| image | octaves | input | time/image | Mpix/s | keypoints/s | keypoints | device time | bytes_alg / device time | octave-0 blur |")
run(512, "white", 0, 50)
run(2048, "white", 0, 30)
run(2048, "smooth", 0, 20)
run(4096, "white", 3, 20)
run(4096, "white", 0, 20)
run(4096, "white", 3, 10, device_input=False)
run(4096, "smooth", 0, 10)
if "--skip-16k" not in sys.argv:
    run(16384, "white", 0, 3)
    run(16384, "white", 3, 3)
# C4 on one GPU: 64 images 2048x2048: frame-by-frame SiftPlan loop, then the pipelined BatchPlan (8 lanes)
imgs = [np.random.default_rng(i).random((2048, 2048), dtype=np.float32) for i in range(64)]
dev = [torch.from_numpy(i).cuda() for i in imgs]
plan = sp.SiftPlan(shape=(2048, 2048), dtype=np.float32)
keypoints_batch(dev[:4], plan=plan)
torch.cuda.synchronize(); t0 = time.perf_counter()
res = keypoints_batch(dev, plan=plan)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("C4 on ONE GPU, SiftPlan loop : 64 x 2048^2 device-resident: %.2f ms total, %.3f ms/image, %.0f Mpix/s, %d keypoints" % (1e3 * dt, 1e3 * dt / 64, 64 * 2048 * 2048 / 1e6 / dt, sum(len(r) for r in res)))
del plan
bp = sp.BatchPlan(shape=(2048, 2048), dtype=np.float32, lanes=8)
bp.keypoints_batch(dev[:8])
for name, frames in (("device-resident", dev), ("host (numpy)", imgs)):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    res = bp.keypoints_batch(frames)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("C4 on ONE GPU, BatchPlan(8)  : 64 x 2048^2 %s: %.2f ms total, %.3f ms/image, %.0f Mpix/s, %d keypoints" % (name, 1e3 * dt, 1e3 * dt / 64, 64 * 2048 * 2048 / 1e6 / dt, sum(len(r) for r in res)))
