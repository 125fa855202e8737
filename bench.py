#!/usr/bin/env python
"""bench.py -- SiftPlan.keypoints() throughput on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step = one SiftPlan.keypoints() over one 4096x4096 fp32 synthetic image (uniform white noise, the
BASELINE.json input) that is already resident in HBM when the timed region starts; 3 octaves x 3
scales (BASELINE.json configs[1]).  With N > 1 each rank (one process per GPU) processes its own
images -- independent units, no data-path collective -- and the timed region ends with the one
exchange step of the batched path, an RCCL all-gather of the keypoint records of the last step.
Rank 0 prints ONE JSON line.

roofline: the dominant kernel family is the fused separable Gaussian blur (blur_march_kernel<N,NORM>,
16 launches per image, 42.7 % of the GPU time in profiles/r01/rocprofv3_summary.txt).  Its algorithmic traffic is
1 read + 1 write of the plane = 8 B per pixel per launch ("5 chained blurs: 5R + 5W"); achieved =
6 * 8 * W*H / (hipEvent time around the six full-resolution launches of an image: initial blur + the five scales
of octave 0), measured live with HIP events on the plan's own pyramid stream; those launches run alone on the GPU,
the later octaves' launches overlap the detection streams and cannot be timed in isolation.
roofline_pipeline uses the whole-call model bytes_alg = W*H*(12 + 66*sum_o 4^-o) + 144 B/keypoint
over the hipEvent time of all kernels of a call.

cpu_baseline: the CPU oracle (a port of the reference's OpenCL-CPU kernels, OpenMP over all host
cores) timed on rank 0 on the same workload -- a reported baseline, not the target.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
SIZE = 4096
OCTAVES = 3


def make_image(seed, size=SIZE):
    return np.random.default_rng(seed).random((size, size), dtype=np.float32)


def bytes_alg(w, h, n_oct, n_kp):
    return w * h * (12.0 + 66.0 * sum(4.0 ** -o for o in range(n_oct))) + 144.0 * n_kp


def cpu_baseline(size, octaves):
    """Oracle (port of the reference CPU kernels) on the host cores; bounded sample."""
    from oracle import pyoracle
    threads = os.cpu_count() or 1
    os.environ.setdefault("OMP_NUM_THREADS", str(threads))
    img = make_image(0, size)
    par = pyoracle.default_params(octave_max=octaves)
    pyoracle.keypoints(make_image(1, 512), par)          # load + thread pool warm-up
    t0 = time.perf_counter()
    reps = 0
    nkp = 0
    while True:
        k = pyoracle.keypoints(img, par)
        nkp = len(k)
        reps += 1
        el = time.perf_counter() - t0
        if el > 10.0 or reps >= 16:
            break
    mpix = reps * size * size / 1e6 / el
    return {"value": round(mpix, 3), "unit": "Mpix/s", "cores": threads, "kind": "port",
            "keypoints_per_s": round(reps * nkp / el, 1),
            "sample": "%d x SiftPlan-equivalent pass over one %dx%d fp32 white-noise image, %d octaves, "
                      "oracle/sift_oracle.c with OpenMP on %d threads (%.1f s wall)" % (reps, size, size, octaves, threads, el)}


def cpu_reference_kernels(octaves, size=1024):
    """The reference's OWN OpenCL-CPU kernels, compiled natively into oracle/_ref (when that build travelled with the
    snapshot), driven serially on one host thread over a bounded sample.  None if the library is absent."""
    try:
        from oracle import pyref
        if not pyref.available():
            return None
        img = make_image(0, size)
        pyref.keypoints(make_image(1, 256), octave_max=octaves)
        t0 = time.perf_counter()
        reps = 0
        while True:
            k = pyref.keypoints(img, octave_max=octaves)
            reps += 1
            el = time.perf_counter() - t0
            if el > 4.0 or reps >= 4:
                break
        return {"value": round(reps * size * size / 1e6 / el, 3), "unit": "Mpix/s", "cores": 1, "kind": "reference",
                "keypoints_per_s": round(reps * len(k) / el, 1),
                "sample": "%d x the reference's own kernels (openCL/*.cl built natively, oracle/_ref) over one %dx%d fp32 "
                          "white-noise image, %d octaves, serial NDRange on 1 thread (%.1f s wall)" % (reps, size, size, octaves, el)}
    except Exception as exc:          # the baseline is a report, never a reason to lose the bench line
        return {"error": str(exc)[:200]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--size", type=int, default=SIZE, help="image side (default: the BASELINE config, 4096)")
    ap.add_argument("--octaves", type=int, default=OCTAVES, help="0 = every octave (reference default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pipelined", action="store_true", help="skip the extra BatchPlan measurement")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse the N>1 "
                                                      "code path on a box with fewer GPUs than ranks)")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal only: every rank uses cuda:0")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device is visible (there is no CPU fallback)")
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    distributed = world > 1
    # collectives run on device tensors with RCCL; the gloo rehearsal stages them through the host
    xdev = "cuda" if args.backend == "nccl" else "cpu"
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=args.backend)

    import sift_pyocl_amd as sp
    from sift_pyocl_amd.batch import RECORD_BYTES

    size, K, W = args.size, args.steps, args.warmup
    plan = sp.SiftPlan(shape=(size, size), dtype=np.float32, devicetype="GPU", device=local_rank, profile="light",
                       octave_max=args.octaves or None)
    n_oct = plan.octave_max
    # inputs resident in HBM before the timed region (distinct images, seeds as SURVEY 8d)
    n_img = min(max(K, 1), 8)
    dev_images = [torch.from_numpy(make_image(rank * 1000 + i, size)).cuda() for i in range(n_img)]
    torch.cuda.synchronize()

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    last = None
    for i in range(W):
        last = plan.keypoints(dev_images[i % n_img])
    if distributed:   # warm the collective path too
        t = torch.zeros(8, dtype=torch.uint8, device=xdev)
        o = torch.empty(8 * world, dtype=torch.uint8, device=xdev)
        dist.all_gather_into_tensor(o, t)

    blur_ms = blur_px = tot_ms = 0.0
    blur_launches = 0
    b0_ms = b0_px = 0.0
    b0_launches = 0
    n_kp = 0
    barrier()
    t0 = time.perf_counter()
    for i in range(K):
        last = plan.keypoints(dev_images[i % n_img])
        n_kp += len(last)
        kt = plan.kernel_times()
        blur_ms += kt["blur_ms"]; blur_px += kt["blur_pixels"]; blur_launches += kt["blur_launches"]
        tot_ms += kt["total_ms"]
        b0_ms += kt["blur0_ms"]; b0_px += kt["blur0_pixels"]; b0_launches += kt["blur0_launches"]
    if distributed:
        # the batched path's single exchange step: all-gather of the keypoint records (padded)
        cnt = torch.tensor([len(last)], dtype=torch.int64, device=xdev)
        cnts = torch.empty(world, dtype=torch.int64, device=xdev)
        dist.all_gather_into_tensor(cnts, cnt)
        mx = int(cnts.max().item())
        buf = torch.zeros(max(1, mx) * RECORD_BYTES, dtype=torch.uint8, device=xdev)
        raw = torch.from_numpy(np.ascontiguousarray(last).view(np.uint8).reshape(-1).copy())
        buf[:raw.numel()] = raw.to(xdev)
        allbuf = torch.empty(world * buf.numel(), dtype=torch.uint8, device=xdev)
        dist.all_gather_into_tensor(allbuf, buf)
    barrier()
    elapsed = time.perf_counter() - t0
    if distributed:
        el = torch.tensor([elapsed], dtype=torch.float64, device=xdev)
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        elapsed = float(el.item())
        tk = torch.tensor([float(n_kp)], dtype=torch.float64, device=xdev)
        dist.all_reduce(tk, op=dist.ReduceOp.SUM)
        total_kp = float(tk.item())
    else:
        total_kp = float(n_kp)

    # Extra, outside the timed region and never part of `value`: the pipelined path (BatchPlan) on the same
    # frames -- what a caller with a stack of frames gets from one GPU (N = 1 only).
    pipelined = None
    if world == 1 and not args.no_pipelined:
        try:
            del plan
            frames = [dev_images[i % n_img] for i in range(16)]
            bp = sp.BatchPlan(shape=(size, size), dtype=np.float32, device=local_rank, octave_max=args.octaves or None, lanes=2)
            bp.keypoints_batch(frames)                      # warm-up: sizes the result arena
            torch.cuda.synchronize()
            times = []
            for _ in range(4):
                t1 = time.perf_counter()
                res = bp.keypoints_batch(frames)
                torch.cuda.synchronize()
                times.append(time.perf_counter() - t1)
            sys.stderr.write("[bench] pipelined call times (ms): %s\n" % ", ".join("%.2f" % (1e3 * t) for t in times))
            tb = sorted(times)[len(times) // 2]
            pipelined = {"value": round(len(frames) * size * size / 1e6 / tb, 2), "unit": "Mpix/s", "ms_per_frame": round(1e3 * tb / len(frames), 4),
                         "frames_per_call": len(frames), "lanes": 2, "keypoints": int(sum(len(r) for r in res)),
                         "note": "BatchPlan.keypoints_batch: frames pipelined over 2 plans, one result copy; every frame "
                                 "bit-identical to SiftPlan.keypoints (tests/test_gpu_batch.py)"}
            del bp
        except Exception as exc:
            pipelined = {"error": str(exc)[:200]}

    if rank == 0:
        mpix_total = world * K * size * size / 1e6
        value = mpix_total / elapsed
        blur_gbs = (8.0 * b0_px / 1e9) / (b0_ms / 1e3) if b0_ms > 0 else 0.0          # full-resolution launches
        blur_all_gbs = (8.0 * blur_px / 1e9) / (blur_ms / 1e3) if blur_ms > 0 else 0.0
        # HBM traffic per blur launch from the committed rocprofv3 PMC passes of this same command
        # (FETCH_SIZE x2 + WRITE_SIZE, see tools/summarize_prof.py); null when absent or another config
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "blur_traffic.json")
        if os.path.exists(tfile) and size == SIZE and n_oct == OCTAVES:
            try:
                traffic = round(json.load(open(tfile))["traffic_bytes_per_launch"], 1)
            except Exception:
                traffic = None
        kp_per_img = n_kp / max(K, 1)
        pipe_gbs = (bytes_alg(size, size, n_oct, kp_per_img) * K / 1e9) / (tot_ms / 1e3) if tot_ms > 0 else 0.0
        out = {
            "metric": "SiftPlan.keypoints throughput, 4096x4096 fp32 (Mpix/s; keypoints/s in keypoints_per_s)",
            "value": round(value, 2), "unit": "Mpix/s",
            "keypoints_per_s": round(total_kp / elapsed, 1),
            "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": round(1e3 * elapsed / max(K, 1), 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "SiftPlan %dx%d fp32 uniform white noise (numpy default_rng(seed).random), "
                                   "%d octaves x 3 scales, input resident in HBM, records returned to host"
                                   % (size, size, n_oct),
                       "octaves": n_oct, "scales": 3, "keypoints_per_image": round(kp_per_img, 1),
                       "images_per_gpu_per_step": 1, "exchange": "rccl all_gather of keypoint records" if distributed else "none"},
            "roofline": {"bound": "hbm",
                         "kernel": "blur_march_kernel<N,NORM>, the %d full-resolution (octave 0) launches per image: initial "
                                   "blur + 5 scales (79 %% of all blur bytes at 3 octaves); they never overlap another kernel, "
                                   "later octaves run concurrently with the detection streams" % (b0_launches // max(K, 1)),
                         "achieved": round(blur_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(blur_gbs / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "avg_launch_us": round(1e3 * b0_ms / max(b0_launches, 1), 2),
                         "alg_bytes_per_launch": round(8.0 * b0_px / max(b0_launches, 1), 1),
                         "timing": "one hipEvent pair on the plan's pyramid stream around the 6 back-to-back launches "
                                   "(inter-kernel gaps included)"},
            "roofline_pipeline": {"bound": "hbm", "achieved": round(pipe_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": round(pipe_gbs / HBM_PEAK_GBS, 4),
                                  "kernel_ms_per_image": round(tot_ms / max(K, 1), 4),
                                  "bytes_alg_per_image": bytes_alg(size, size, n_oct, kp_per_img)},
        }
        if pipelined is not None:
            out["pipelined"] = pipelined
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(size, n_oct)
            ref = cpu_reference_kernels(n_oct)
            if ref is not None:
                out["cpu_baseline_reference_kernels"] = ref
        print(json.dumps(out), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
