#!/usr/bin/env python
"""MatchPlan timing (BASELINE.json configs[4]: 100k x 100k 128-D uint8 descriptors, L1 + ratio test as the
reference).  Reports kernel time (hipEvent), end-to-end time with host lists and with device-resident lists,
and the byte-SAD rate against the v_sad_u8 issue bound."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sift_pyocl_amd as sp

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
dtype_kp = sp.MatchPlan.dtype_kp
rng = np.random.default_rng(1)
a = np.zeros(n, dtype_kp); a["desc"] = rng.integers(0, 256, (n, 128), dtype=np.uint8)
rng2 = np.random.default_rng(2)
b = np.zeros(n, dtype_kp)
perm = rng2.permutation(n); half = n // 2
b["desc"][:half] = np.clip(a["desc"][perm[:half]].astype(np.int16) + rng2.integers(-8, 9, (half, 128)), 0, 255).astype(np.uint8)
b["desc"][half:] = rng2.integers(0, 256, (n - half, 128), dtype=np.uint8)
mp = sp.MatchPlan(size=n)
for _ in range(2):
    pairs = mp.match(a, b, raw_results=True)
t0 = time.perf_counter(); pairs = mp.match(a, b, raw_results=True); t_host = time.perf_counter() - t0
ta = torch.from_numpy(a.view(np.uint8).reshape(-1)).cuda(); tb = torch.from_numpy(b.view(np.uint8).reshape(-1)).cuda()
torch.cuda.synchronize()
mp.match(ta, tb, raw_results=True)
t0 = time.perf_counter(); pairs_d = mp.match(ta, tb, raw_results=True); t_dev = time.perf_counter() - t0
ms = mp.kernel_ms()
sads = float(n) * n * 128
print("n=%d pairs=%d (expected %d)  kernel %.3f ms  e2e host lists %.3f ms  device lists %.3f ms" % (n, len(pairs), half, ms, 1e3 * t_host, 1e3 * t_dev))
print("byte-SADs/s %.3e  (v_sad_u8 issue bound ~3.1e14: %.1f %%)  pairs/s %.3e" % (sads / (ms / 1e3), 100 * sads / (ms / 1e3) / 3.1e14, float(n) * n / (ms / 1e3)))
