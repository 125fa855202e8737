#!/usr/bin/env python
"""LinearAlign on one MI355X: warp-kernel roofline and end-to-end align() time.

    python tools/bench_align.py [--size 4096] [--reps 10]

Warp kernel: algorithmic bytes = 1 read + 1 write of the plane = 8 B/pixel (float32), 6 B/pixel (RGB8).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=4096)
    ap.add_argument("--reps", type=int, default=10)
    a = ap.parse_args()
    import sift_pyocl_amd as sp
    from sift_pyocl_amd import _lib
    from scipy.ndimage import gaussian_filter
    S = a.size
    rng = np.random.default_rng(0)
    big = gaussian_filter(rng.random((S + 64, S + 64), dtype=np.float32), 2.0).astype(np.float32)
    ref = np.ascontiguousarray(big[20:20 + S, 30:30 + S]); img = np.ascontiguousarray(big[27:27 + S, 19:19 + S])
    la = sp.LinearAlign(ref)
    out = {"size": S, "ref_keypoints": int(len(la.ref_kp))}
    for name, kw in (("shift_only", dict(shift_only=True)), ("affine", dict())):
        la.align(img, **kw)
        t0 = time.perf_counter()
        for _ in range(a.reps):
            la.align(img, **kw)
        dt = (time.perf_counter() - t0) / a.reps
        r = la.align(img, return_all=True, **kw)
        out[name] = {"align_ms": round(1e3 * dt, 3), "matches": int(r["matching"].shape[0]), "offset": [float(v) for v in r["offset"]],
                     "rms": float(r["rms"]), "transform_kernel_ms": round(la.last_transform_ms, 4),
                     "sift_kernel_ms": round(la.sift.kernel_times()["total_ms"], 3) if la.sift.profile else None,
                     "match_kernel_ms": round(la.match.kernel_ms(), 3)}
    # warp kernel alone, device-resident output, small affine
    L = _lib.lib()
    M = np.array([0.999, 0.012, -0.011, 1.002], np.float32); off = np.array([5.3, -7.9], np.float32)
    import torch
    dout = torch.empty((S, S), dtype=torch.float32, device="cuda")
    din = torch.from_numpy(img).cuda()
    ms = C.c_double()
    times = []
    for i in range(a.reps + 3):
        _lib.check(L.siftmi_plan_transform(la.sift._handle, din.data_ptr(), 1, 1, dout.data_ptr(), 1, S, S, M.ctypes.data, off.ctypes.data,
                                           C.c_float(0.0), 1, C.byref(ms)))
        if i >= 3:
            times.append(ms.value)
    k_ms = float(np.mean(times))
    out["transform_kernel"] = {"ms": round(k_ms, 4), "alg_bytes": 8.0 * S * S, "GBps": round(8.0 * S * S / 1e9 / (k_ms / 1e3), 1),
                               "frac_of_8TBps": round(8.0 * S * S / 1e9 / (k_ms / 1e3) / 8000.0, 4)}
    rgb = rng.integers(0, 256, (S, S, 3), dtype=np.uint8)
    pr = sp.SiftPlan(template=rgb, octave_max=1)
    din = torch.from_numpy(rgb).cuda(); dout = torch.empty((S, S, 3), dtype=torch.uint8, device="cuda")
    times = []
    for i in range(a.reps + 3):
        _lib.check(L.siftmi_plan_transform(pr._handle, din.data_ptr(), 1, 3, dout.data_ptr(), 1, S, S, M.ctypes.data, off.ctypes.data,
                                           C.c_float(0.0), 1, C.byref(ms)))
        if i >= 3:
            times.append(ms.value)
    k_ms = float(np.mean(times))
    out["transform_rgb_kernel"] = {"ms": round(k_ms, 4), "alg_bytes": 6.0 * S * S, "GBps": round(6.0 * S * S / 1e9 / (k_ms / 1e3), 1)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
