"""GPU parity at the BASELINE.json sizes, plus the committed golden vectors replayed through the
HIP path.  The oracle (OpenMP) finishes a 4096x4096 image in about a second on the GPU box's host
cores, so full-size parity is checked directly, bit for bit, in addition to size-independent
properties (determinism of the keypoint set, bounds, plan reuse, record/descriptor consistency)."""
import os

import numpy as np
import pytest

from util import (assert_same_keypoints, compare_keypoints_libm, digest_cases, kp_digest, multiscale_noise, rectangles,
                  smooth_noise, sort_kp, sort_rows, white_noise)

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("name,maker,shape", [("white512", white_noise, (512, 512)), ("smooth512", smooth_noise, (512, 512)),
                                              ("multi300x421", multiscale_noise, (300, 421)),
                                              ("rect257x511", rectangles, (257, 511))])
def test_golden_final_keypoints(siftlib, name, maker, shape):
    """HIP output against vectors produced by the reference's own kernels with glibc math: tolerance of
    compare_keypoints_libm (on these four small images no descriptor bin differs at all)."""
    import sift_pyocl_amd as sp
    g = np.load(os.path.join(GOLD, "kp_%s.npz" % name))
    img = maker(shape)
    got = sp.SiftPlan(template=img).keypoints(img)
    stats = compare_keypoints_libm(got, g["kp"], name)
    assert stats["desc_bins_differing"] == 0


@pytest.mark.parametrize("name", sorted(digest_cases()))
def test_golden_digests_of_large_images(siftlib, name):
    """HIP output == the reference's own kernels (math builtins bound to siftmath, oracle/_ref/libsiftclref_sm.so) on
    2048 x 2048 / 1031 x 1537 frames, every byte of 39 k / 2.7 k / 18.7 k records, through committed per-field digests
    (tests/golden/kp_digests.json, generator tests/golden/make_golden.py; needs neither the oracle nor the reference)."""
    import json
    import sift_pyocl_amd as sp
    maker, shape, kw = digest_cases()[name]
    img = maker(shape, **kw)
    golden = json.load(open(os.path.join(GOLD, "kp_digests.json")))[name]
    plan = sp.SiftPlan(template=img)
    assert kp_digest(plan.keypoints(img)) == golden
    # the second call of a plan on a keypoint-rich frame switches to full gradient maps by itself (option "maps" = 2);
    # forced on and off as well: same bytes
    assert kp_digest(plan.keypoints(img)) == golden
    for maps in (1, 0):
        plan.set_option("maps", maps)
        assert kp_digest(plan.keypoints(img)) == golden, "maps=%d" % maps


def test_4096_white_noise_bit_exact(siftlib, oracle):
    """BASELINE.json configs[1]: 4096x4096 fp32, 3 octaves x 3 scales -- and all 9 octaves."""
    import sift_pyocl_amd as sp
    img = white_noise((4096, 4096), seed=0)
    plan3 = sp.SiftPlan(template=img, octave_max=3)
    got3 = plan3.keypoints(img)
    assert_same_keypoints(got3, oracle.keypoints(img, oracle.default_params(octave_max=3)), "4096 white, 3 octaves")
    plan = sp.SiftPlan(template=img)
    assert plan.octave_max == 9 and plan.kpsize == 4096 * 4096 // 10
    got = plan.keypoints(img)
    assert_same_keypoints(got, oracle.keypoints(img), "4096 white, all octaves")
    assert 9000 < len(got) < 13000 and not plan.overflow
    # properties: inside the image, positive scale, angles in [-pi, pi], set is reproducible
    assert (got.x >= 0).all() and (got.x < 4096).all() and (got.y >= 0).all() and (got.y < 4096).all()
    assert (got.scale > 1.0).all() and (np.abs(got.angle) <= np.float32(np.pi)).all()
    assert_same_keypoints(got, plan.keypoints(img), "second call on the same plan")
    sub = {r.tobytes() for r in got3}
    assert sub.issubset({r.tobytes() for r in got}), "3-octave result must be a subset of the full one"


def test_2048_keypoint_rich_bit_exact(siftlib, oracle):
    import sift_pyocl_amd as sp
    img = smooth_noise((2048, 2048), seed=3)
    plan = sp.SiftPlan(template=img)
    got = plan.keypoints(img)
    assert len(got) > 30000
    want = oracle.keypoints(img)
    assert_same_keypoints(got, want, "2048 smoothed noise")
    # launch-layout options that only large frames reach (none may change a byte): the later octaves as one chain and group or
    # as two, one stream -- each with the lazy gradient and with full maps
    # ... the tiles in plain workgroup order (xcd_map = 0), the octaves below octave 1 searched on a stream of their own into
    # the one group of the later octaves (split; with maps = 1 it does not apply and the plain single chain runs)
    for opts in (dict(fork=0), dict(fork=1), dict(early_chain=0), dict(early_chain=1, fork=1), dict(overlap=0), dict(xcd_map=0),
                 dict(split=1, fork=0), dict(split=1, fork=0, early_chain=1)):
        for maps in (0, 1):
            p2 = sp.SiftPlan(template=img)
            p2.set_option("maps", maps)
            for k, v in opts.items():
                p2.set_option(k, v)
            assert_same_keypoints(p2.keypoints(img), want, "2048 smoothed noise, %r maps=%d" % (opts, maps))
            assert_same_keypoints(p2.keypoints(img), want, "2048 smoothed noise, %r maps=%d, second call" % (opts, maps))


def test_odd_sizes_and_borders(siftlib, oracle):
    import sift_pyocl_amd as sp
    for shape in [(1031, 1537), (2050, 1026), (1024, 4100)]:
        img = smooth_noise(shape, seed=shape[0])
        got = sp.SiftPlan(template=img).keypoints(img)
        assert_same_keypoints(got, oracle.keypoints(img), str(shape))


def test_capacity_overflow_is_reported(siftlib):
    import sift_pyocl_amd as sp
    img = smooth_noise((512, 512))
    plan = sp.SiftPlan(template=img, PIX_PER_KP=1000)      # kpsize 262 per octave << ~1950 keypoints in octave 0
    got = plan.keypoints(img)
    assert plan.overflow and len(got) <= plan.octave_max * plan.kpsize      # (tests/test_gpu_capacity.py has the rule in full)


def test_match_100k_against_the_full_oracle(siftlib, oracle):
    """BASELINE.json configs[4] (100k x 100k): properties of the planted matches, then EVERY pair against the OpenMP
    oracle run in full when the host has the cores for it (1.28e12 byte differences: seconds on the GPU box's 256
    threads); on a small host a 300-query slice is compared instead."""
    import os
    import sift_pyocl_amd as sp
    from util import dtype_kp
    n = 100000
    rng = np.random.default_rng(1)
    a = np.zeros(n, dtype_kp); a["desc"] = rng.integers(0, 256, (n, 128), dtype=np.uint8)
    rng2 = np.random.default_rng(2)
    b = np.zeros(n, dtype_kp)
    perm = rng2.permutation(n)
    half = n // 2
    b["desc"][:half] = np.clip(a["desc"][perm[:half]].astype(np.int16) + rng2.integers(-8, 9, (half, 128)), 0, 255).astype(np.uint8)
    b["desc"][half:] = rng2.integers(0, 256, (n - half, 128), dtype=np.uint8)
    b["desc"][half + 5] = b["desc"][7]      # an exact duplicate of a planted match: best == second best, the ratio test drops it
    mp = sp.MatchPlan()
    pairs = mp.match(a, b, raw_results=True)
    assert len(pairs) == half - 1                               # every other perturbed copy matches, random ones never do
    inv = np.empty(n, np.int64); inv[perm[:half]] = np.arange(half)
    assert (pairs[:, 1] == inv[pairs[:, 0]]).all() and perm[7] not in pairs[:, 0]
    assert len(np.unique(pairs[:, 0])) == half - 1
    if (os.cpu_count() or 1) >= 64:
        exp, total = oracle.match(a, b, cap=n)
        assert total == len(pairs) and np.array_equal(sort_rows(pairs), sort_rows(exp))
    else:
        sl = np.sort(rng.choice(n, 300, replace=False))
        exp, total = oracle.match(a[sl], b)
        got = pairs[np.isin(pairs[:, 0], sl)].copy()
        got[:, 0] = np.searchsorted(sl, got[:, 0])
        assert total == len(got) and np.array_equal(sort_rows(got), sort_rows(exp))


def test_16384_white_noise_bit_exact(siftlib, oracle):
    """BASELINE.json configs[2]: 16384 x 16384 fp32 on one MI355X (1 GiB frame, 8.5 GB of planes): bit-exact against
    the oracle at 3 octaves (the oracle needs ~20 s on the host cores), plus bounds on the full-octave run."""
    import sift_pyocl_amd as sp
    img = white_noise((16384, 16384), seed=2)
    plan = sp.SiftPlan(template=img, octave_max=3)
    got = plan.keypoints(img)
    assert 150000 < len(got) < 200000 and not plan.overflow
    assert_same_keypoints(got, oracle.keypoints(img, oracle.default_params(octave_max=3)), "16384 white, 3 octaves")
    del plan
    full = sp.SiftPlan(template=img)
    assert full.octave_max == 11
    allk = full.keypoints(img)
    assert {r.tobytes() for r in got}.issubset({r.tobytes() for r in allk})
