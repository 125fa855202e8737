"""dev: interleaved A/B of plan option sets on one frame (medians over alternating rounds, same process and box):
   python tools/dev/ab_opts.py "desc_blocks=2048" "desc_blocks=768" "desc_pad=10000,split_detect=0" [size=4096] [octaves=3] [kind=white]"""
import os, sys, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import sift_pyocl_amd as sp
from util import smooth_noise
sets = [a for a in sys.argv[1:] if not a.startswith(("size=", "octaves=", "kind=", "rounds=", "profile="))]
kw = dict(a.split("=") for a in sys.argv[1:] if a.startswith(("size=", "octaves=", "kind=", "rounds=", "profile=")))
size = int(kw.get("size", 4096)); octaves = int(kw.get("octaves", 3)); rounds = int(kw.get("rounds", 15))
img = smooth_noise((size, size)) if kw.get("kind") == "smooth" else np.random.default_rng(0).random((size, size), dtype=np.float32)
t = torch.from_numpy(img).cuda()
plans = []
for s in sets:
    plan = sp.SiftPlan(shape=img.shape, dtype=np.float32, octave_max=octaves or None, **({"profile": kw["profile"]} if "profile" in kw else {}))
    for kv in s.split(","):
        if "=" in kv and kv != "base=1":
            name, v = kv.split("="); plan.set_option(name, int(v))
    for _ in range(5): plan.keypoints(t)
    plans.append(plan)
times = [[] for _ in plans]
for r in range(rounds):
    for i, plan in enumerate(plans):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): plan.keypoints(t)
        times[i].append((time.perf_counter() - t0) / 10)
for s, ts in zip(sets, times):
    print("%-48s median %.4f ms  min %.4f  max %.4f" % (s, 1e3 * statistics.median(ts), 1e3 * min(ts), 1e3 * max(ts)), flush=True)
