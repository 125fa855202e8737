"""dev: interleaved A/B of a 0 / 1 plan option (default: xcd_map, the workgroup -> XCD mapping of the tiled kernels): whole call and the light profile's
bracket around the six full-resolution blur launches, records compared between the two plans.
   python tools/dev/ab_xcd.py [opt=xcd_map] [fixed=fork:0,...] [size=4096] [octaves=3] [rounds=15]"""
import os, sys, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import sift_pyocl_amd as sp
kw = dict(a.split("=") for a in sys.argv[1:])
size = int(kw.get("size", 4096)); octaves = int(kw.get("octaves", 3)); rounds = int(kw.get("rounds", 15)); inner = int(kw.get("inner", 10))
opt = kw.get("opt", "xcd_map")
fixed = [kv.split(":") for kv in kw.get("fixed", "").split(",") if kv]      # fixed=fork:0,early_chain:0
from util import smooth_noise
img = smooth_noise((size, size)) if kw.get("kind") == "smooth" else np.random.default_rng(0).random((size, size), dtype=np.float32)
t = torch.from_numpy(img).cuda()
plans, recs = [], []
vals = [int(x) for x in kw.get("vals", "0,1").split(",")]
for v in vals:
    plan = sp.SiftPlan(shape=img.shape, dtype=np.float32, octave_max=octaves or None, profile="light")
    for name, val in fixed: plan.set_option(name, int(val))
    plan.set_option(opt, v)
    for _ in range(5): kp = plan.keypoints(t)
    recs.append(np.sort(np.frombuffer(np.ascontiguousarray(kp).tobytes(), dtype="S144")))
    plans.append(plan)
print("records: %s, identical: %s" % ([len(r) for r in recs], all(len(r) == len(recs[0]) and bool((r == recs[0]).all()) for r in recs)), flush=True)
times = [[] for _ in vals]; blur = [[] for _ in vals]
for r in range(rounds):
    for i, plan in enumerate(plans):
        plan.profile_totals(reset=True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(inner): plan.keypoints(t)
        times[i].append((time.perf_counter() - t0) / inner)
        tot = plan.profile_totals(reset=True)
        blur[i].append(tot["blur0_ms"] / max(tot["blur0_launches"], 1))
for i in range(len(vals)):
    print("%s=%d  call median %.4f ms (min %.4f)   blur launch median %.2f us (min %.2f)" % (
        opt, vals[i], 1e3 * statistics.median(times[i]), 1e3 * min(times[i]), 1e3 * statistics.median(blur[i]), 1e3 * min(blur[i])), flush=True)
