"""GPU parity for the LinearAlign row: the affine warp kernel through the C ABI against
the oracle and the golden vectors of the reference's transform.cl, and LinearAlign.align end to end."""
import ctypes as C
import os

import numpy as np
import pytest

from util import TRANSFORM_CASES, smooth_noise, transform_inputs, white_noise

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def biteq(a, b):
    a = np.ascontiguousarray(a); b = np.ascontiguousarray(b)
    return a.shape == b.shape and a.dtype == b.dtype and np.array_equal(a.view(np.uint8), b.view(np.uint8))


def warp(siftlib, plan, image, M, off, out_shape, fill, mode):
    """straight through the C ABI (include/siftmi.h: siftmi_plan_transform)"""
    from sift_pyocl_amd import _lib
    rgb = image.ndim == 3
    image = np.ascontiguousarray(image, np.uint8 if rgb else np.float32)
    out = np.empty(tuple(out_shape) + ((3,) if rgb else ()), image.dtype)
    M = np.ascontiguousarray(M, np.float32).reshape(4); off = np.ascontiguousarray(off, np.float32).reshape(2)
    ms = C.c_double()
    _lib.check(siftlib.siftmi_plan_transform(plan._handle, image.ctypes.data, 0, 3 if rgb else 1, out.ctypes.data, 0,
                                             out_shape[1], out_shape[0], M.ctypes.data, off.ctypes.data, C.c_float(fill), mode,
                                             C.byref(ms)))
    return out


def test_transform_golden(siftlib):
    import sift_pyocl_amd as sp
    g = np.load(os.path.join(GOLD, "transform.npz"))
    gray, rgb = transform_inputs()
    pg = sp.SiftPlan(template=gray); pr = sp.SiftPlan(template=rgb)
    for i, (M, off, fill, mode, extra) in enumerate(TRANSFORM_CASES):
        og = tuple(s + e for s, e in zip(gray.shape, extra or (0, 0)))
        orgb = tuple(s + e for s, e in zip(rgb.shape[:2], extra or (0, 0)))
        assert biteq(warp(siftlib, pg, gray, M, off, og, fill, mode), g["gray%d" % i]), "gray case %d" % i
        assert biteq(warp(siftlib, pr, rgb, M, off, orgb, fill, mode), g["rgb%d" % i]), "rgb case %d" % i


@pytest.mark.parametrize("shape", [(1, 1), (3, 70), (129, 65), (500, 777), (2048, 2048)])
def test_transform_random_affine_vs_oracle(siftlib, oracle, shape):
    import sift_pyocl_amd as sp
    rng = np.random.default_rng(shape[0] * 7 + shape[1])
    img = (white_noise(shape, seed=5) * 500.0 - 100.0).astype(np.float32)
    plan = sp.SiftPlan(shape=shape, dtype=np.float32) if min(shape) > 12 else None
    if plan is None:
        pytest.skip("SiftPlan needs at least one octave; the warp is bound to a plan")
    for k in range(6):
        M = (np.eye(2) + rng.normal(0, 0.05 if k else 0.0, (2, 2))).astype(np.float32).reshape(4)
        off = rng.normal(0, 0.02 * max(shape), 2).astype(np.float32)
        oshape = shape if k % 2 == 0 else (shape[0] + 9, shape[1] + 30)
        for mode in (1, 0):
            got = warp(siftlib, plan, img, M, off, oshape, 13.0, mode)
            assert biteq(got, oracle.transform(img, M, off, out_shape=oshape, fill=13.0, mode=mode)), (k, mode)


def test_transform_rgb_vs_oracle(siftlib, oracle):
    import sift_pyocl_amd as sp
    rng = np.random.default_rng(9)
    rgb = rng.integers(0, 256, (301, 403, 3), dtype=np.uint8)
    plan = sp.SiftPlan(template=rgb)
    for k in range(5):
        M = (np.eye(2) + rng.normal(0, 0.05, (2, 2))).astype(np.float32).reshape(4)
        off = rng.normal(0, 8, 2).astype(np.float32)
        assert biteq(warp(siftlib, plan, rgb, M, off, (301, 403), 0.0, 1), oracle.transform(rgb, M, off, fill=0.0, mode=1))


def test_transform_4096_full_size(siftlib, oracle):
    """BASELINE-size plane: bit-exact against the oracle, plus size-independent properties."""
    import sift_pyocl_amd as sp
    img = white_noise((4096, 4096), seed=1)
    plan = sp.SiftPlan(template=img, octave_max=3)
    M = np.array([0.999, 0.012, -0.011, 1.002], np.float32); off = np.array([5.3, -7.9], np.float32)
    got = warp(siftlib, plan, img, M, off, (4096, 4096), -1.0, 1)
    assert biteq(got, oracle.transform(img, M, off, fill=-1.0, mode=1))
    # integer shift = exact copy of the shifted block; everything that fell outside = fill
    got = warp(siftlib, plan, img, [1, 0, 0, 1], [16, -32], (4096, 4096), -1.0, 1)
    assert np.array_equal(got[:4079, 32:], img[16:4095, :4064])
    assert (got[:, :32] == -1.0).all() and (got[4080:, :] == -1.0).all()


def test_transform_argument_errors(siftlib):
    import sift_pyocl_amd as sp
    from sift_pyocl_amd import _lib
    img = smooth_noise((64, 64))
    plan = sp.SiftPlan(template=img)
    out = np.empty((64, 64), np.float32); M = np.eye(2, dtype=np.float32); off = np.zeros(2, np.float32)
    # nothing staged yet -> image=NULL is an error, not a read of garbage
    rc = siftlib.siftmi_plan_transform(plan._handle, None, 0, 1, out.ctypes.data, 0, 64, 64, M.ctypes.data, off.ctypes.data, C.c_float(0), 1, None)
    assert rc == _lib.EINVAL
    rc = siftlib.siftmi_plan_transform(plan._handle, img.ctypes.data, 0, 2, out.ctypes.data, 0, 64, 64, M.ctypes.data, off.ctypes.data, C.c_float(0), 1, None)
    assert rc == _lib.EINVAL
    plan.keypoints(img)                       # stages a float32 image
    rc = siftlib.siftmi_plan_transform(plan._handle, None, 0, 3, out.ctypes.data, 0, 64, 64, M.ctypes.data, off.ctypes.data, C.c_float(0), 1, None)
    assert rc == _lib.EINVAL                  # staged image is float32, RGB requested
    rc = siftlib.siftmi_plan_transform(plan._handle, None, 0, 1, out.ctypes.data, 0, 64, 64, M.ctypes.data, off.ctypes.data, C.c_float(0), 1, None)
    assert rc == 0 and np.array_equal(out[:63, :63], img[:63, :63])


def test_align_shift_only_recovers_translation(siftlib, oracle):
    import sift_pyocl_amd as sp
    big = smooth_noise((600, 640), seed=12, sigma=2.0)
    ref_img = np.ascontiguousarray(big[20:532, 30:542])
    img = np.ascontiguousarray(big[27:539, 19:531])          # content moved by dy=-7, dx=+11 in array coordinates
    la = sp.LinearAlign(ref_img)
    res = la.align(img, shift_only=True, return_all=True)
    assert res["matching"].shape[0] > 100
    assert np.array_equal(res["matrix"], np.identity(2, dtype=np.float32))
    assert abs(res["offset"][0] - (-7.0)) < 0.05 and abs(res["offset"][1] - 11.0) < 0.05
    assert res["rms"] < 0.5
    # the warp applied is the oracle's, bit for bit, for the transformation LinearAlign derived
    want = oracle.transform(img, res["matrix"].reshape(4), res["offset"], fill=float(img.min()), mode=1)
    assert biteq(res["result"], want)
    # and the aligned image matches the reference where both are defined
    inner = (slice(20, 480), slice(20, 480))
    assert np.abs(res["result"][inner] - ref_img[inner]).max() < 0.02 * (ref_img.max() - ref_img.min())


def test_align_affine_matches_cpu_pipeline(siftlib, oracle):
    """Full align() against the same pipeline assembled from the oracle (keypoints, match, warp) and the numpy glue."""
    import sift_pyocl_amd as sp
    from sift_pyocl_amd.utils import matching_correction
    from util import dtype_kp, sort_kp
    ref_img = smooth_noise((480, 512), seed=14, sigma=2.0)
    M_true = np.array([1.004, 0.018, -0.017, 0.997], np.float32); off_true = np.array([3.4, -2.2], np.float32)
    img = oracle.transform(ref_img, M_true, off_true, fill=0.0, mode=1)
    la = sp.LinearAlign(ref_img)
    res = la.align(img, return_all=True)
    # CPU restatement of align(): order-independent because both keypoint lists are sorted first
    k_ref = sort_kp(oracle.keypoints(ref_img)); kp = sort_kp(oracle.keypoints(img))
    assert np.array_equal(sort_kp(la.ref_kp).view(np.uint8), k_ref.view(np.uint8))
    assert np.array_equal(sort_kp(res["keypoint"]).view(np.uint8), kp.view(np.uint8))
    pairs, n = oracle.match(k_ref, kp)
    assert n == res["matching"].shape[0] and n >= 18
    m = np.recarray(shape=(n, 2), dtype=dtype_kp)
    m[:, 0] = k_ref[pairs[:n, 0]]; m[:, 1] = kp[pairs[:n, 1]]
    got_pairs = {(a.tobytes(), b.tobytes()) for a, b in zip(res["matching"][:, 0], res["matching"][:, 1])}
    assert got_pairs == {(a.tobytes(), b.tobytes()) for a, b in zip(m[:, 0], m[:, 1])}
    # the solved transformation depends on the (unordered) pair list only through a least-squares sum
    t = matching_correction(m)
    assert np.allclose([t[4], t[3], t[1], t[0]], res["matrix"].reshape(4), rtol=0, atol=2e-6)
    assert np.allclose([t[5], t[2]], res["offset"], rtol=0, atol=2e-4)
    want = oracle.transform(img, res["matrix"].reshape(4), res["offset"], fill=float(img.min()), mode=1)
    assert biteq(res["result"], want)
    assert res["rms"] < 0.3
    inner = (slice(30, 450), slice(30, 480))
    assert np.abs(res["result"][inner] - ref_img[inner]).mean() < 0.01 * (ref_img.max() - ref_img.min())


def test_align_rgb_and_extra_and_relative(siftlib, oracle):
    import sift_pyocl_amd as sp
    rng = np.random.default_rng(3)
    base = smooth_noise((300, 330), seed=16, sigma=1.5)
    base = (255 * (base - base.min()) / (base.max() - base.min()))
    rgb_big = np.stack([base, base[::-1, ::-1], base.T[:300, :300].repeat(2, axis=1)[:, :330]], axis=-1).astype(np.uint8)
    ref_img = np.ascontiguousarray(rgb_big[10:266, 12:268]); img = np.ascontiguousarray(rgb_big[14:270, 9:265])
    la = sp.LinearAlign(ref_img, extra=(4, 6))
    assert la.RGB and la.outshape == (264, 268)
    res = la.align(img, shift_only=True, return_all=True)
    assert res["result"].shape == (264, 268, 3) and res["result"].dtype == np.uint8
    want = oracle.transform(img, res["matrix"].reshape(4), res["offset"], out_shape=la.outshape, fill=float(la.sift.minmax()[0]), mode=1)
    assert biteq(res["result"], want)
    assert abs(res["offset"][0] + 4.0) < 0.1 and abs(res["offset"][1] - 3.0) < 0.1
    # relative mode chains the transformations and replaces the reference keypoints
    la2 = sp.LinearAlign(ref_img)
    r1 = la2.align(img, shift_only=True, relative=True, return_all=True)
    assert la2.relative_transfo is not None and len(la2.ref_kp) == len(r1["keypoint"])
    r2 = la2.align(img, shift_only=True, relative=True, return_all=True)
    assert np.allclose(r2["offset"], r1["offset"], atol=1e-3)        # second step is the identity shift
    # no keypoints at all -> None, as the reference
    flat = np.zeros_like(ref_img)
    assert la.align(flat) is None


def test_align_log_profile_lists_its_own_and_its_plans_events(siftlib, oracle, capsys):
    """LinearAlign(profile=True).log_profile() (alignment.py:363-375): every entry is (label, event) with the reference's
    ``1e-6 * (evt.profile.end - evt.profile.start)`` reading; the matcher's stages (match.py:226-263) are among them."""
    import sift_pyocl_amd as sp
    big = smooth_noise((420, 460), seed=21, sigma=2.0)
    ref_img = np.ascontiguousarray(big[10:394, 12:396]); img = np.ascontiguousarray(big[14:398, 9:393])
    la = sp.LinearAlign(ref_img, profile=True)
    plain = sp.LinearAlign(ref_img)
    res = la.align(img, shift_only=True, return_all=True)
    assert biteq(res["result"], plain.align(img, shift_only=True))              # profiling does not change the result
    assert [l for l, _ in la.events] == ["transform"]
    assert "matching" in [l for l, _ in la.match.events] and any("descriptors" in l for l, _ in la.sift.events)
    for label, evt in la.events + la.match.events + la.sift.events:
        assert 0 <= evt.profile.end - evt.profile.start < 1e9, label
    la.log_profile()
    printed = capsys.readouterr().out
    for needle in ("transform", "matching", "copy D->H match", "descriptors", "Total execution time"):
        assert needle in printed, needle
    plain.log_profile()
    assert capsys.readouterr().out == ""
