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
    # ---- second image, two octaves: the facts only this repository's launch sequencer pinned until round 5 -- octsize = 2
    # (the reference's own fixture uses it, test/test_image_setup.py:24-81): the EdgeThresh0 / EdgeThresh slot order of
    # local_maxmin (is_maxmin picks by octsize, test_image_functions.py:85; plan.py:633-634 passes EdgeThresh1 first), the
    # x = col * octsize scaling of orientation and descriptor -- on enough keypoints (> 500 descriptors).  The planes come from
    # the oracle's blur (pinned bit for bit against the reference's convolution kernels, tests/test_oracle_golden.py): the numpy
    # functions have no pyramid of their own beyond my_blur, and every stage below it is the reference's numpy code alone.
    from oracle import pyoracle
    from util import multiscale_noise, oracle_pyramid
    img2 = multiscale_noise((300, 421))
    for o, (blurs2, dogs2) in enumerate(oracle_pyramid(pyoracle, img2, 2)):
        octsize = 1 << o
        H2, W2 = dogs2.shape[1:]
        for s in (1, 2, 3):
            tag = "m_o%d_s%d_" % (o, s)
            cand, n = ns["my_local_maxmin"](dogs2, peakthresh, 5, octsize, np.float32(0.08), np.float32(0.06), 20000, s, W2, H2)
            cand = cand[:n]
            out[tag + "candidates"] = cand
            ref = np.array([ns["my_interp_keypoint"](dogs2, s, int(k[1]), int(k[2]), 5, peakthresh, W2, H2) for k in cand], np.float32).reshape(-1, 4)
            out[tag + "interp"] = ref
            grad, ori = ns["my_gradient"](blurs2[s])
            kin = ref[ref[:, 1] != -1]                     # the numpy-refined keypoints feed the numpy orientation
            nb = len(kin)
            buf = -np.ones((4 * nb + 64, 4), np.float32)
            buf[:nb] = kin
            okp, cnt = ns["my_orientation"](buf, len(buf), 0, nb, grad, ori, octsize, np.float32(1.5))
            out[tag + "oriented"] = okp[:cnt].copy()
            out[tag + "desc"] = ns["my_descriptor"](okp[:cnt].copy(), grad, ori, octsize, 0, cnt)
            print(tag, "candidates", n, "refined", nb, "oriented", cnt, flush=True)
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
