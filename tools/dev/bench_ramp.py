"""dev: is the slowness of the first dozen calls after an idle period a device clock ramp?  20 steps (after 5 warm-up)
timed right after (a) nothing, (b) 100 ms of a compute-bound torch kernel, (c) 100 ms of the workload itself."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import sift_pyocl_amd as sp
size = 4096
img = torch.from_numpy(np.random.default_rng(0).random((size, size), dtype=np.float32)).cuda()
plan = sp.SiftPlan(shape=(size, size), dtype=np.float32, octave_max=3)
a = torch.randn(4096, 4096, device="cuda"); b = torch.randn(4096, 4096, device="cuda")
def measure(label, prep):
    time.sleep(1.0)
    prep()
    for _ in range(5): plan.keypoints(img)
    torch.cuda.synchronize(); ts = []
    for _ in range(20):
        t1 = time.perf_counter(); plan.keypoints(img); ts.append(1e3 * (time.perf_counter() - t1))
    print("%-28s mean %.4f  first5 %.4f  last5 %.4f" % (label, sum(ts) / 20, sum(ts[:5]) / 5, sum(ts[-5:]) / 5), flush=True)
def nothing(): pass
def matmul():
    t_end = time.perf_counter() + 0.1
    while time.perf_counter() < t_end:
        for _ in range(4): torch.mm(a, b)
        torch.cuda.synchronize()
def workload():
    t_end = time.perf_counter() + 0.1
    while time.perf_counter() < t_end: plan.keypoints(img)
for rep in range(2):
    measure("after 1 s idle", nothing)
    measure("after 100 ms of fp32 matmul", matmul)
    measure("after 100 ms of the workload", workload)
