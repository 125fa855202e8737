"""The oracle against the reference's OWN numpy restatements (test/test_image_functions.py:13-480), through outputs only:
tests/golden/numpy_xcheck.npz was produced by tests/golden/make_numpy_xcheck.py, which exec's that file where it lies in
the build container -- no OpenCL shim and no launch sequencer of this repository sits under those numbers.

The numpy functions are float64 and deliberately looser than the kernels, so this is a semantic cross-check (window and
histogram definitions, launch order, thresholds), not a bit-level one; tolerances are the reference's own
(test/test_image.py:128-129 gradient 1e-4, :189-191 extrema 1e-4, :252 interpolation 1e-4; test/test_keypoints.py:306-309
orientation 1e-4 / 1e-4 / 1e-4 / 1e-1; descriptors: "several difference of 1", test_keypoints.py:300-304).
"""
import os

import numpy as np
import pytest

from util import sort_rows

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def data():
    return np.load(os.path.join(HERE, "golden", "stages_131x97.npz")), np.load(os.path.join(HERE, "golden", "numpy_xcheck.npz"))


def by_position(a):
    return a[np.lexsort((a[:, 2], a[:, 1]))]


@pytest.mark.parametrize("s", [1, 2, 3])
def test_local_maxmin_equals_the_numpy_restatement(oracle, data, s):
    z, x = data
    kps, n = oracle.local_maxmin(z["o0_dogs"], s, 1, 4096)
    got, want = by_position(kps[:n]), by_position(x["s%d_candidates" % s])
    assert got.shape == want.shape                       # same candidate set (no borderline `<` / `<=` case on this image)
    assert np.array_equal(got[:, 1:], want[:, 1:])       # row, column, scale
    assert np.abs(got[:, 0] - want[:, 0]).max() < 1e-4   # peak value (test_image.py:189)


@pytest.mark.parametrize("s", [1, 2, 3])
def test_interp_keypoint_against_the_numpy_restatement(oracle, data, s):
    z, x = data
    cand = x["s%d_candidates" % s]
    got = oracle.interp_keypoint(z["o0_dogs"], cand, 0, len(cand))
    want = x["s%d_interp" % s]
    gv, wv = got[:, 1] != -1, want[:, 1] != -1
    # accept / reject: the numpy version tests abs(x) < 1.5 and peak > thresh where the kernel has <= and >=, and its move
    # loop compares with the ORIGINAL position (test_image_functions.py:130-139): rows may differ only on such borderline
    # candidates -- none does on this image
    assert np.array_equal(gv, wv), "accept/reject differs at rows %r" % (np.nonzero(gv != wv)[0],)
    d = np.abs(got[gv] - want[wv]).max(axis=1)
    bad = np.nonzero(d >= 1e-4)[0]                       # test_image.py:252 (float64 inverse against the kernel's f32 adjugate)
    # One candidate of the 97 on this image (scale 1, row 45, column 88) oscillates between two pixels: the kernel moves
    # `while (moves > 0 && moved)` (image.cl:335-349) and reports the last position it evaluated, the numpy loop stops as soon
    # as the walk returns to the ORIGINAL pixel (test_image_functions.py:135).  Same pixel pair, different end of the walk.
    assert len(bad) <= (1 if s == 1 else 0), "rows %r differ" % (bad,)
    for i in bad:
        assert np.abs(got[gv][i, 1:3] - want[wv][i, 1:3]).max() < 1.0 and abs(got[gv][i, 0] - want[wv][i, 0]) < 0.5


@pytest.mark.parametrize("s", [1, 2, 3])
def test_gradient_against_numpy_gradient(oracle, data, s):
    z, x = data
    g, o = oracle.gradient(z["o0_blurs"][s])
    assert np.abs(g - x["s%d_grad" % s]).max() < 1e-4    # test_image.py:128
    d = np.abs(o - x["s%d_ori" % s])
    d = np.minimum(d, 2 * np.pi - d)                     # +pi and -pi are the same direction (zero vertical gradient, negative horizontal)
    flat = g < 1e-3                                      # the angle of a (nearly) zero gradient is noise in either implementation
    assert d[~flat].max() < 1e-4 and flat.mean() < 0.01  # test_image.py:129


@pytest.mark.parametrize("s", [1, 2, 3])
def test_orientation_against_the_numpy_restatement(oracle, data, s):
    z, x = data
    kin = z["o0_s%d_refined" % s]
    nb = len(kin)
    buf = -np.ones((4 * nb + 64, 4), np.float32)
    buf[:nb] = kin
    g, o = oracle.gradient(z["o0_blurs"][s])
    okp, cnt = oracle.orientation(buf, g, o, 1, 0, nb)
    got, want = okp[:cnt], x["s%d_oriented" % s]
    assert got.shape == want.shape                       # same number of additional orientations
    # the reference's comparison: every column sorted on its own (test_image_functions.py:435-448, test_keypoints.py:306-309)
    for col, tol in ((0, 1e-4), (1, 1e-4), (2, 1e-4), (3, 1e-1)):
        assert np.abs(np.sort(got[:, col]) - np.sort(want[:, col])).max() < tol
    # ... and the stronger, matched one: same (x, y, sigma) rows, angles within the binning noise of float32 against float64
    # (a sample on a bin edge of orientation_cpu.cl:88 moves a peak by a fraction of a bin = 10 degrees)
    key = lambda a: a[np.lexsort((a[:, 3], a[:, 2], a[:, 1], a[:, 0]))]   # noqa: E731
    gs, ws = key(got), key(want)
    assert np.abs(gs[:, :3] - ws[:, :3]).max() < 1e-4
    assert np.abs(gs[:, 3] - ws[:, 3]).max() < 1e-5          # measured: 4.8e-7 (132 oriented keypoints)


@pytest.mark.parametrize("s", [1, 2, 3])
def test_descriptor_against_the_numpy_restatement(oracle, data, s):
    z, x = data
    okp = x["s%d_oriented" % s]                          # the numpy-oriented keypoints, so that only the descriptor differs
    g, o = oracle.gradient(z["o0_blurs"][s])
    got = oracle.descriptor(okp, g, o, 1, 0, len(okp)).astype(int)
    want = x["s%d_desc" % s].astype(int)
    assert got.shape == want.shape
    diff = np.abs(got - want)
    # "several difference of 1" (test_keypoints.py:300): float32 accumulation in raster order against float64; the numpy
    # version also clamps with (vec > 0.2) on EVERY descriptor and renormalises all of them if any was clamped
    # measured on the 132 descriptors of this image: every one of the 16 896 bins is EQUAL
    assert diff.max() <= 1, "largest bin difference %d" % diff.max()
    assert (diff > 0).mean() < 0.01


def test_matching_against_the_numpy_restatement(oracle, data):
    import sys
    sys.path.insert(0, os.path.join(HERE, "golden"))
    from make_golden import match_sets
    _, x = data
    a, b = match_sets()
    na, nb = [int(v) for v in x["match_sizes"]]
    pairs, total = oracle.match(a[:na], b[:nb])
    got = np.asarray(pairs[:total], np.int64).reshape(-1, 2)
    want = x["match_pairs"]
    # my_matching keeps a pair only if i <= match (test_image_functions.py:392; not in matching_cpu.cl): compare on that subset
    got_sub = got[got[:, 0] <= got[:, 1]]
    assert len(want) > 5 and np.array_equal(sort_rows(got_sub), sort_rows(want))


# ---- second image (multiscale noise, 300 x 421), octaves 0 and 1: octsize = 2 through every stage -- the edge-threshold slot
# order of local_maxmin (plan.py:633-634 passes EdgeThresh1 then EdgeThresh, the kernel picks by octsize, image.cl:193), the
# x = col * octsize scaling of the oriented keypoints and the descriptor's division by octsize (keypoints_cpu.cl:64-65) -- on 900
# descriptors.  Planes: the oracle's own (oracle_pyramid; its blur is pinned bit for bit against the reference's convolution
# kernels elsewhere); everything the comparison is about is the reference's numpy code alone.
@pytest.fixture(scope="module")
def second(oracle):
    from util import multiscale_noise, oracle_pyramid
    return oracle_pyramid(oracle, multiscale_noise((300, 421)), 2)


M_CASES = [(o, s) for o in (0, 1) for s in (1, 2, 3)]


@pytest.mark.parametrize("o,s", M_CASES)
def test_second_image_local_maxmin(oracle, data, second, o, s):
    _, x = data
    dogs = second[o][1]
    kps, n = oracle.local_maxmin(dogs, s, 1 << o, 20000)
    got, want = by_position(kps[:n]), by_position(x["m_o%d_s%d_candidates" % (o, s)])
    assert len(want) > 25
    # float64 against float32 in the edge test (det < edthresh * tr^2) and the contrast pre-test: a candidate within a few
    # ulp of a threshold may fall on either side -- count them, none is expected to be far from its threshold
    gk = set(map(tuple, got[:, 1:].astype(int).tolist())); wk = set(map(tuple, want[:, 1:].astype(int).tolist()))
    assert not (gk ^ wk), "candidate sets differ: %r" % (sorted(gk ^ wk)[:5],)        # measured: all 764 candidates of the six cases equal
    both = sorted(gk & wk)
    gd = {tuple(r[1:].astype(int)): r[0] for r in got}; wd = {tuple(r[1:].astype(int)): r[0] for r in want}
    assert max(abs(gd[k] - wd[k]) for k in both) < 1e-4          # peak value (test_image.py:189)


@pytest.mark.parametrize("o,s", M_CASES)
def test_second_image_interp_keypoint(oracle, data, second, o, s):
    _, x = data
    dogs = second[o][1]
    cand = x["m_o%d_s%d_candidates" % (o, s)]
    got = oracle.interp_keypoint(dogs, cand, 0, len(cand))
    want = x["m_o%d_s%d_interp" % (o, s)]
    gv, wv = got[:, 1] != -1, want[:, 1] != -1
    # borderline accept / reject (abs(x) < 1.5 against <= 1.5f, peak > against >=) and walks that end on the other pixel of an
    # oscillating pair (see the first image's test): a handful of rows at most
    flips = int((gv != wv).sum())
    assert flips == 0, "accept / reject differs in %d of %d rows" % (flips, len(cand))    # measured: none of the 764
    both = gv & wv
    d = np.abs(got[both] - want[both]).max(axis=1)
    bad = np.nonzero(d >= 1e-4)[0]
    assert len(bad) <= 1, "%d of %d refined rows differ by more than 1e-4" % (len(bad), int(both.sum()))   # measured: none (an oscillating walk would be one)
    for i in bad:                                                    # ... and those are walks between neighbouring samples
        assert np.abs(got[both][i, 1:3] - want[both][i, 1:3]).max() < 1.5


@pytest.mark.parametrize("o,s", M_CASES)
def test_second_image_orientation(oracle, data, second, o, s):
    _, x = data
    blurs = second[o][0]
    ref = x["m_o%d_s%d_interp" % (o, s)]
    kin = ref[ref[:, 1] != -1]
    nb = len(kin)
    buf = -np.ones((4 * nb + 64, 4), np.float32)
    buf[:nb] = kin
    g, ori = oracle.gradient(blurs[s])
    okp, cnt = oracle.orientation(buf, g, ori, 1 << o, 0, nb)
    got, want = okp[:cnt], x["m_o%d_s%d_oriented" % (o, s)]
    # a histogram peak at exactly 80 % of the maximum, or a sample on a bin edge, may add or drop an extra orientation
    assert len(got) == len(want), "%d against %d oriented keypoints" % (len(got), len(want))       # measured: 900 of 900
    key = lambda a: a[np.lexsort((a[:, 3], a[:, 2], a[:, 1], a[:, 0]))]   # noqa: E731
    if len(got) == len(want):
        gs, ws = key(got), key(want)
        assert np.abs(gs[:, :3] - ws[:, :3]).max() < 1e-3 * (1 << o)       # x, y, sigma: col * octsize, row * octsize, sigma * octsize
        assert np.abs(gs[:, 3] - ws[:, 3]).max() < 1e-3                     # measured: every angle (the reference's own bound: 1e-1 on the sorted column)
    # positions are multiples of octsize times the refined (row, col): the scaling itself
    assert np.allclose(np.sort(np.unique(got[:, 0])), np.sort(np.unique(kin[:, 2] * (1 << o))), atol=1e-3)


@pytest.mark.parametrize("o,s", M_CASES)
def test_second_image_descriptor(oracle, data, second, o, s):
    _, x = data
    blurs = second[o][0]
    okp = x["m_o%d_s%d_oriented" % (o, s)]                 # the numpy-oriented keypoints, so that only the descriptor differs
    g, ori = oracle.gradient(blurs[s])
    got = oracle.descriptor(okp, g, ori, 1 << o, 0, len(okp)).astype(int)
    want = x["m_o%d_s%d_desc" % (o, s)].astype(int)
    assert got.shape == want.shape and len(want) >= 40
    diff = np.abs(got - want)
    # "several difference of 1" (test_keypoints.py:300): float32 accumulation in raster order against float64
    # measured over the 900 descriptors of the six cases: ONE of 115 200 bins differs, by 1
    assert diff.max() <= 1 and int((diff > 0).sum()) <= 3, "%d bins differ, largest difference %d" % (int((diff > 0).sum()), diff.max())
