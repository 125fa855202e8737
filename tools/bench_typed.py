#!/usr/bin/env python
"""Typed frames (u8 / u16 / RGB8) through SiftPlan.keypoints: fused converters vs the convert pass.

    python tools/bench_typed.py [--size 4096] [--reps 20]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=4096)
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    import torch
    import sift_pyocl_amd as sp
    S = a.size
    rng = np.random.default_rng(0)
    out = {"size": S, "octaves": 3}
    frames = {"float32": rng.random((S, S), dtype=np.float32),
              "uint8": rng.integers(0, 256, (S, S), dtype=np.uint8),
              "uint16": rng.integers(0, 65536, (S, S), dtype=np.uint16),
              "rgb8": rng.integers(0, 256, (S, S, 3), dtype=np.uint8)}
    for name, img in frames.items():
        plan = sp.SiftPlan(template=img, octave_max=3)
        t = torch.from_numpy(img).cuda()
        res = {}
        for mode in ("fused", "convert_pass"):
            if name == "float32" and mode == "convert_pass":
                continue
            plan.set_option("fused_convert", 0 if mode == "convert_pass" else 1)
            for _ in range(3):
                k = plan.keypoints(t)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.reps):
                k = plan.keypoints(t)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / a.reps
            res[mode] = {"ms": round(1e3 * dt, 4), "Mpix_s": round(S * S / 1e6 / dt, 1), "keypoints": int(len(k))}
        out[name] = res
    print(json.dumps(out))


if __name__ == "__main__":
    main()
