"""dev: what the bench loop adds to a bare keypoints() loop on the headline frame (profile brackets, kernel_times, frame rotation)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import sift_pyocl_amd as sp
size = 4096
imgs = [torch.from_numpy(np.random.default_rng(i).random((size, size), dtype=np.float32)).cuda() for i in range(8)]
def run(label, profile, times, rotate):
    plan = sp.SiftPlan(shape=(size, size), dtype=np.float32, octave_max=3, profile=profile)
    for i in range(5): plan.keypoints(imgs[0])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(30):
        k = plan.keypoints(imgs[i % 8 if rotate else 0])
        if times: plan.kernel_times()
    torch.cuda.synchronize()
    print("%-40s %.4f ms" % (label, 1e3 * (time.perf_counter() - t0) / 30), flush=True)
run("bare", False, False, False)
run("rotate frames", False, False, True)
run("light profile", "light", False, False)
run("light profile + kernel_times", "light", True, False)
run("light + times + rotate (= bench)", "light", True, True)
run("bare", False, False, False)
