#!/usr/bin/env python
"""Throughput with several SiftPlans driven from concurrent host threads (each plan has its own streams
and buffers; ctypes releases the GIL during the call).  python tools/concurrent_plans.py [nplans] [steps]"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sift_pyocl_amd as sp

nplans = int(sys.argv[1]) if len(sys.argv) > 1 else 2
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 24
size = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
imgs = [torch.from_numpy(np.random.default_rng(i).random((size, size), dtype=np.float32)).cuda() for i in range(8)]
torch.cuda.synchronize()
plans = [sp.SiftPlan(shape=(size, size), dtype=np.float32, octave_max=(3 if size >= 4096 else None)) for _ in range(nplans)]
for p in plans:
    for i in range(2): p.keypoints(imgs[i])
counts = [0] * nplans
def work(k):
    for i in range(k, steps, nplans):
        counts[k] += len(plans[k].keypoints(imgs[i % 8]))
torch.cuda.synchronize()
t0 = time.perf_counter()
th = [threading.Thread(target=work, args=(k,)) for k in range(nplans)]
for t in th: t.start()
for t in th: t.join()
torch.cuda.synchronize()
el = time.perf_counter() - t0
print("plans=%d steps=%d  %.3f ms/image  %.0f Mpix/s  %.0f keypoints/s" % (nplans, steps, 1e3 * el / steps, steps * size * size / 1e6 / el, sum(counts) / el))
