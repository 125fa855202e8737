"""dev: the first calls after an idle period are slower than the steady state -- whose idle time is it?
python tools/dev/ramp.py   (ms per call in groups of 5 [blur launch us] after: sleep, host busy-wait, GPU busy + host busy)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import sift_pyocl_amd as sp
from bench import make_image
size = 4096
plan = sp.SiftPlan(shape=(size, size), dtype=np.float32, devicetype="GPU", profile="light", octave_max=3)
imgs = [torch.from_numpy(make_image(i, size)).cuda() for i in range(5)]
torch.cuda.synchronize()

def groups(tag, n=10):
    out = []
    for g in range(n):
        plan.profile_totals(reset=True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(5):
            plan.keypoints(imgs[i])
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5 * 1e3
        t = plan.profile_totals(reset=True)
        out.append("%.3f[%.1f]" % (dt, 1e3 * t["blur0_ms"] / max(t["blur0_launches"], 1)))
    print(tag, " ".join(out), flush=True)

def spin(sec):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < sec:
        pass

groups("first use            :")
for rep in range(2):
    time.sleep(2.0);  groups("after sleep 2 s      :")
    spin(2.0);        groups("after host spin 2 s  :")
    a = torch.randn(4096, 4096, device="cuda")
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 2.0:
        b = a @ a
        torch.cuda.synchronize()
    groups("after matmul+sync 2 s:")
    time.sleep(0.05); groups("after sleep 50 ms    :")
    spin(0.05);       groups("after spin 50 ms     :")
