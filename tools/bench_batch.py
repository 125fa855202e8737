#!/usr/bin/env python
"""Batched, pipelined keypoints (SURVEY 8f-4 / BASELINE.json configs[3]) against the frame-by-frame SiftPlan loop.

    python tools/bench_batch.py [--size 2048] [--frames 64] [--lanes 1,2,4,8] [--octaves 0]
Frames are device resident (torch tensors); times are wall clock around whole calls.
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
    ap.add_argument("--size", type=int, default=2048)
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--lanes", default="1,2,4,8")
    ap.add_argument("--octaves", type=int, default=0)
    a = ap.parse_args()
    import torch
    import sift_pyocl_amd as sp
    S, F = a.size, a.frames
    n_distinct = min(F, 8)
    host = [np.random.default_rng(i).random((S, S), dtype=np.float32) for i in range(n_distinct)]
    dev = [torch.from_numpy(h).cuda() for h in host]
    frames = [dev[i % n_distinct] for i in range(F)]
    torch.cuda.synchronize()
    out = {"size": S, "frames": F, "octaves": a.octaves or "all"}
    plan = sp.SiftPlan(shape=(S, S), dtype=np.float32, octave_max=a.octaves or None)
    for f in frames[:3]:
        plan.keypoints(f)
    t0 = time.perf_counter()
    nk = sum(len(plan.keypoints(f)) for f in frames)
    dt = time.perf_counter() - t0
    out["siftplan_loop"] = {"ms_per_frame": round(1e3 * dt / F, 4), "Mpix_s": round(F * S * S / 1e6 / dt, 1), "keypoints": nk}
    del plan
    for lanes in [int(x) for x in a.lanes.split(",")]:
        bp = sp.BatchPlan(shape=(S, S), dtype=np.float32, lanes=lanes, octave_max=a.octaves or None)
        bp.keypoints_batch(frames[:max(lanes, 2)])
        t0 = time.perf_counter()
        res = bp.keypoints_batch(frames)
        dt = time.perf_counter() - t0
        out["batch_lanes_%d" % lanes] = {"ms_per_frame": round(1e3 * dt / F, 4), "Mpix_s": round(F * S * S / 1e6 / dt, 1),
                                         "keypoints": int(sum(len(r) for r in res)), "plan_GB": round(bp.memory / 1e9, 2)}
        del bp
    print(json.dumps(out))


if __name__ == "__main__":
    main()
