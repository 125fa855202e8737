#!/usr/bin/env python
"""Cross-check fixture from the reference's OWN numpy restatements (test/test_image_functions.py:13-480): the one
reference-authored statement of the per-stage semantics that can be executed in the build container without any code
of this repository underneath it (no OpenCL shim, no launch sequencer).

Runs only where /root/reference is mounted.  The reference file is read where it lies, tab-expanded (it mixes tabs and
spaces, test_image_functions.py:407-411) and exec'd in-process; it never enters this repository or the GPU box.  Only
OUTPUTS are stored: tests/golden/numpy_xcheck.npz.  Inputs are the arrays already committed in stages_131x97.npz (the
DoG / blur planes of octave 0 of the 131x97 smoothed-noise image) and the two descriptor lists of match.npz's generator.

    python tests/golden/make_numpy_xcheck.py

The functions are float64 and looser than the kernels (`abs(x) < 1.5` against `<= 1.5f`, `peakval > thr` against `>=`,
test_image_functions.py:139 against image.cl:351; the loop test of my_interp_keypoint compares with the ORIGINAL position):
tests/test_oracle_numpy_xcheck.py compares the oracle with these outputs at the tolerances of the reference's own tests
(test/test_image.py:128-129, 189-191, 252; test/test_keypoints.py:306-309) and counts the rows that differ.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
REF = "/root/reference/test/test_image_functions.py"

from make_golden import match_sets  # noqa: E402


def load_reference_functions():
    text = open(REF).read().expandtabs(8)
    ns = {}
    exec(compile(text, REF, "exec"), ns)
    return ns


def main():
    ns = load_reference_functions()
    z = np.load(os.path.join(HERE, "stages_131x97.npz"))
    dogs, blurs = z["o0_dogs"], z["o0_blurs"]
    H, W = dogs.shape[1:]
    peakthresh = np.float32(255.0 * 0.04 / 3.0)             # plan.py:631
    out = {}
    for s in (1, 2, 3):
        cand, n = ns["my_local_maxmin"](dogs, peakthresh, 5, 1, np.float32(0.08), np.float32(0.06), 4096, s, W, H)
        cand = cand[:n]
        out["s%d_candidates" % s] = cand
        ref = np.array([ns["my_interp_keypoint"](dogs, s, int(k[1]), int(k[2]), 5, peakthresh, W, H) for k in cand], np.float32).reshape(-1, 4)
        out["s%d_interp" % s] = ref
        grad, ori = ns["my_gradient"](blurs[s])
        out["s%d_grad" % s] = np.asarray(grad, np.float32)
        out["s%d_ori" % s] = np.asarray(ori, np.float32)
        # orientation + descriptors from the REFERENCE-KERNEL refined list of the committed fixture (the same input the
        # oracle is given in the test), with the numpy gradient maps
        kin = z["o0_s%d_refined" % s]
        nb = len(kin)
        buf = -np.ones((4 * nb + 64, 4), np.float32)
        buf[:nb] = kin
        okp, cnt = ns["my_orientation"](buf, len(buf), 0, nb, grad, ori, 1, np.float32(1.5))
        out["s%d_oriented" % s] = okp[:cnt].copy()
        ndesc = cnt
        out["s%d_desc" % s] = ns["my_descriptor"](okp[:ndesc].copy(), grad, ori, 1, 0, ndesc)
    a, b = match_sets()
    na, nb_ = 300, 240
    da, db = a["desc"][:na].astype(np.int64), b["desc"][:nb_].astype(np.int64)
    pairs, cnt = ns["my_matching"](da, db, 0, na)
    out["match_pairs"] = np.asarray(pairs[:cnt], np.int64)
    out["match_sizes"] = np.array([na, nb_])
    np.savez_compressed(os.path.join(HERE, "numpy_xcheck.npz"), **out)
    for k, v in out.items():
        print(k, v.shape, v.dtype)


if __name__ == "__main__":
    main()
