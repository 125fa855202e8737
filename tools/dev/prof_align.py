import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from scipy.ndimage import gaussian_filter
import sift_pyocl_amd as sp
S = 4096
rng = np.random.default_rng(0)
big = gaussian_filter(rng.random((S + 64, S + 64), dtype=np.float32), 2.0).astype(np.float32)
ref = np.ascontiguousarray(big[20:20 + S, 30:30 + S]); img = np.ascontiguousarray(big[27:27 + S, 19:19 + S])
la = sp.LinearAlign(ref)
for kw in (dict(shift_only=True), dict(), dict(return_all=True)):
    la.align(img, **kw)
    t0 = time.perf_counter()
    for _ in range(3): la.align(img, **kw)
    print(kw, "%.1f ms" % (1e3 * (time.perf_counter() - t0) / 3))
pr = cProfile.Profile(); pr.enable(); la.align(img); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
