"""dev: where LinearAlign.align spends its time (4096^2 smoothed noise, ~200 k keypoints per frame): the phases of align()
timed one by one over several calls."""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from scipy.ndimage import gaussian_filter
import sift_pyocl_amd as sp
S = 4096
rng = np.random.default_rng(0)
big = gaussian_filter(rng.random((S + 64, S + 64), dtype=np.float32), 2.0).astype(np.float32)
ref = np.ascontiguousarray(big[20:20 + S, 30:30 + S]); img = np.ascontiguousarray(big[27:27 + S, 19:19 + S])
la = sp.LinearAlign(ref)
ph = {}
def tick(name, t0):
    ph.setdefault(name, []).append(1e3 * (time.perf_counter() - t0))
for it in range(8):
    t_all = time.perf_counter()
    t0 = time.perf_counter(); kp = la.sift.keypoints(img); tick("keypoints", t0)
    import torch
    t0 = time.perf_counter(); torch.cuda.synchronize(); tick("sync after keypoints", t0)
    t0 = time.perf_counter(); pairs = la.match.match(la._ref_dev, la.sift.device_records(), raw_results=True); tick("match", t0)
    print("  iter %d: match wall %.2f ms, kernel %.2f ms" % (it, ph["match"][-1], la.match.kernel_ms()))
    t0 = time.perf_counter(); g0 = np.ascontiguousarray(la._xysa(la.ref_kp))[pairs[:, 0]]; g1 = np.ascontiguousarray(la._xysa(kp))[pairs[:, 1]]; tick("gather", t0)
    t0 = time.perf_counter(); m, o = la._affine(g0, g1); tick("affine", t0)
    t0 = time.perf_counter(); res = la.transform(m, o, image=None, fill=la.sift.minmax()[0], mode=1); tick("transform", t0)
    tick("sum", t_all)
    t0 = time.perf_counter(); r = la.align(img); tick("align()", t0)
    del res, r
print("keypoints per frame %d, pairs %d" % (len(kp), len(pairs)))
for k, v in ph.items():
    print("%-10s median %.2f ms  min %.2f  max %.2f" % (k, statistics.median(v), min(v), max(v)))
print("match kernel ms", la.match.kernel_ms())
