#!/usr/bin/env python
"""Generate the committed golden vectors from the reference's OWN kernels.

Runs only where /root/reference is mounted: oracle/Makefile compiles the reference's OpenCL C
kernels (openCL/*.cl, where they lie) natively for x86-64 into oracle/_ref/libsiftclref.so, and
oracle/pyref.py sequences them with the launch order and scalars of sift-src/plan.py:432-756 and
sift-src/match.py:200-271.  Nothing from the reference is copied: the outputs below are data
(synthetic inputs -> reference outputs).

    python tests/golden/make_golden.py

Fixtures (all inputs are regenerated from seeds by tests/util.py, only outputs are stored):
  taps.npz            the six Gaussian tap vectors of the default sigma schedule (gaussian.cl)
  stages_131x97.npz   every intermediate of octaves 0 and 1 of a 131x97 smoothed-noise image
  kp_<name>.npz       final keypoints (sorted) of four synthetic images
  match.npz           match pairs of two 1500 / 1200 descriptor sets
  kp_digests.json     per-field SHA-256 of the sorted keypoints of three large images (2048^2 smooth / white,
                      1031x1537), produced by the reference kernels with their math builtins bound to siftmath
                      (oracle/_ref/libsiftclref_sm.so: libm isolated -> the digests are reproducible bit for bit)
  transform.npz       affine warps (transform.cl) of a 97x131 float image and a 40x53 RGB image, cases of
                      tests/util.py:TRANSFORM_CASES
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import pyref  # noqa: E402
from util import digest_cases, kp_digest, TRANSFORM_CASES, transform_inputs, multiscale_noise, rectangles, smooth_noise, sort_kp, sort_rows, white_noise, dtype_kp  # noqa: E402

FINAL_CASES = {"white512": (white_noise, (512, 512)), "smooth512": (smooth_noise, (512, 512)),
               "multi300x421": (multiscale_noise, (300, 421)), "rect257x511": (rectangles, (257, 511))}


def match_sets():
    rng = np.random.default_rng(11)
    a = np.zeros(1500, dtype_kp); b = np.zeros(1200, dtype_kp)
    a["desc"] = rng.integers(0, 256, (1500, 128), dtype=np.uint8)
    b["desc"] = rng.integers(0, 256, (1200, 128), dtype=np.uint8)
    idx = rng.permutation(1500)[:600]
    b["desc"][:600] = np.clip(a["desc"][idx].astype(int) + rng.integers(-8, 9, (600, 128)), 0, 255).astype(np.uint8)
    b["desc"][7] = b["desc"][6]
    a["x"] = np.arange(1500); b["x"] = np.arange(1200)
    return a, b


def main():
    assert pyref.build(), "needs /root/reference to build oracle/_ref"
    first, sig = pyref.sigma_schedule()
    np.savez_compressed(os.path.join(HERE, "taps.npz"), sigmas=np.array([first] + sig),
                        **{"taps%d" % i: pyref.gaussian_taps(s) for i, s in enumerate([first] + sig)})
    img = smooth_noise((131, 97), seed=3, sigma=2.0)
    st = {}
    kp = pyref.keypoints(img, stages=st)
    out = dict(min=st["min"], max=st["max"], base=st["base"], final=sort_kp(kp))
    for o, oc in enumerate(st["octaves"][:2]):
        out["o%d_blurs" % o] = oc["blurs"]
        out["o%d_dogs" % o] = oc["dogs"]
        for sc in oc["scales"]:
            s = sc["scale"]
            out["o%d_s%d_candidates" % (o, s)] = sort_rows(sc["candidates"])
            out["o%d_s%d_interp" % (o, s)] = sc["interp"]          # same order as the unsorted candidates
            out["o%d_s%d_candidates_raw" % (o, s)] = sc["candidates"]
            out["o%d_s%d_refined" % (o, s)] = sc["refined"]
            out["o%d_s%d_oriented" % (o, s)] = sc["oriented"]
            out["o%d_s%d_desc" % (o, s)] = sc["desc"]
            if s == 2:
                out["o%d_s2_grad" % o] = sc["grad"]
                out["o%d_s2_ori" % o] = sc["ori"]
    np.savez_compressed(os.path.join(HERE, "stages_131x97.npz"), **out)
    for name, (maker, shape) in FINAL_CASES.items():
        k = sort_kp(pyref.keypoints(maker(shape)))
        np.savez_compressed(os.path.join(HERE, "kp_%s.npz" % name), kp=k)
        print(name, len(k), "keypoints")
    a, b = match_sets()
    pairs, n = pyref.match(a, b)
    np.savez_compressed(os.path.join(HERE, "match.npz"), pairs=sort_rows(pairs), total=n)
    print("match pairs", n)
    gray, rgb = transform_inputs()
    out = {}
    for i, (M, off, fill, mode, oshape) in enumerate(TRANSFORM_CASES):
        out["gray%d" % i] = pyref.transform(gray, M, off, out_shape=oshape and tuple(s + e for s, e in zip(gray.shape, oshape)), fill=fill, mode=mode)
        out["rgb%d" % i] = pyref.transform(rgb, M, off, out_shape=oshape and tuple(s + e for s, e in zip(rgb.shape[:2], oshape)), fill=fill, mode=mode)
    np.savez_compressed(os.path.join(HERE, "transform.npz"), **out)
    print("transform cases", len(TRANSFORM_CASES))
    import json
    pyref.use("siftmath")
    dig = {}
    for name, (maker, shape, kw) in digest_cases().items():
        dig[name] = kp_digest(pyref.keypoints(maker(shape, **kw)))
        print(name, dig[name]["n"], "keypoints (digest)")
    pyref.use("glibc")
    with open(os.path.join(HERE, "kp_digests.json"), "w") as f:
        json.dump(dig, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
