"""Shared helpers for the test-suite: synthetic inputs and order-insensitive comparison."""
import numpy as np

dtype_kp = np.dtype([("x", np.float32), ("y", np.float32), ("scale", np.float32), ("angle", np.float32),
                     ("desc", (np.uint8, 128))])


def white_noise(shape, seed=0):
    return np.random.default_rng(seed).random(shape, dtype=np.float32)


def smooth_noise(shape, seed=3, sigma=3.0):
    import scipy.ndimage as ndi
    return ndi.gaussian_filter(np.random.default_rng(seed).random(shape), sigma).astype(np.float32)


def multiscale_noise(shape, seed=5):
    import scipy.ndimage as ndi
    rng = np.random.default_rng(seed)
    base = rng.random(shape)
    return sum(ndi.gaussian_filter(base, s) * s for s in (1, 2, 4, 8)).astype(np.float32)


def rectangles(shape, seed=7, n=60):
    rng = np.random.default_rng(seed)
    img = np.zeros(shape, np.float32)
    H, W = shape
    for _ in range(n):
        y0, x0 = rng.integers(0, H - 4), rng.integers(0, W - 4)
        h, w = rng.integers(3, max(4, H // 3)), rng.integers(3, max(4, W // 3))
        img[y0:y0 + h, x0:x0 + w] += rng.random()
    return img


def sort_kp(k):
    """Keypoint order is unspecified (atomic append in the reference, plan.py:3.2): compare sorted."""
    k = np.asarray(k)
    key = np.lexsort((k["desc"][:, 1], k["desc"][:, 0], k["angle"], k["scale"], k["y"], k["x"]))
    return k[key]


def assert_same_keypoints(a, b, what=""):
    """Bit-exact equality of two keypoint sets after sorting."""
    assert len(a) == len(b), "%s: %d vs %d keypoints" % (what, len(a), len(b))
    a, b = sort_kp(a), sort_kp(b)
    for f in ("x", "y", "scale", "angle"):
        fa, fb = a[f].view(np.uint32), b[f].view(np.uint32)
        bad = np.nonzero(fa != fb)[0]
        assert bad.size == 0, "%s: field %s differs at %d rows, e.g. %r vs %r" % (
            what, f, bad.size, a[f][bad[:3]], b[f][bad[:3]])
    bad = np.nonzero((a["desc"] != b["desc"]).any(axis=1))[0]
    assert bad.size == 0, "%s: descriptors differ in %d rows" % (what, bad.size)


def sort_rows(a):
    a = np.asarray(a)
    return a[np.lexsort(a.T[::-1])]


def ulp_diff(a, b):
    a = np.ascontiguousarray(a, np.float32).view(np.int32).astype(np.int64)
    b = np.ascontiguousarray(b, np.float32).view(np.int32).astype(np.int64)
    a = np.where(a < 0, np.int64(-2 ** 31) - a, a)
    b = np.where(b < 0, np.int64(-2 ** 31) - b, b)
    return np.abs(a - b)


def compare_keypoints_libm(a, b, what="", ulp=2, max_bad_rows=0.01):
    """Comparison against the glibc-backed native build of the reference kernels.  x and y (pure
    +,*,/ arithmetic) must be bit-identical; scale (pow) and angle (atan2/exp chain) may differ by
    `ulp` units in the last place because glibc 2.35's powf/atan2f are not correctly rounded while
    the oracle's are; descriptor bins may then differ by 1 LSB in at most `max_bad_rows` of the rows.
    Returns a dict of the measured differences."""
    assert len(a) == len(b), "%s: %d vs %d keypoints" % (what, len(a), len(b))
    a, b = sort_kp(a), sort_kp(b)
    assert np.array_equal(a["x"].view(np.uint32), b["x"].view(np.uint32)), what + ": x differs"
    assert np.array_equal(a["y"].view(np.uint32), b["y"].view(np.uint32)), what + ": y differs"
    ds, da = ulp_diff(a["scale"], b["scale"]), ulp_diff(a["angle"], b["angle"])
    assert ds.max(initial=0) <= ulp, "%s: scale differs by %d ulp" % (what, ds.max())
    assert da.max(initial=0) <= ulp, "%s: angle differs by %d ulp" % (what, da.max())
    dd = np.abs(a["desc"].astype(np.int16) - b["desc"].astype(np.int16))
    assert dd.max(initial=0) <= 1, "%s: descriptor bins differ by %d" % (what, dd.max())
    bad = int(((ds > 0) | (da > 0) | (dd.max(axis=1) > 0)).sum()) if len(a) else 0
    assert bad <= max(2, max_bad_rows * len(a)), "%s: %d of %d rows differ" % (what, bad, len(a))
    return dict(rows=len(a), rows_differing=bad, scale_ulp=int(ds.max(initial=0)), angle_ulp=int(da.max(initial=0)),
                desc_bins_differing=int((dd > 0).sum()))


# (matrix[4], offset[2], fill, mode, extra (dy, dx) added to the output shape or None) -- transform.cl cases
TRANSFORM_CASES = [
    ([1, 0, 0, 1], [0, 0], 0.0, 1, None),                       # identity: last row / column cut to fill
    ([1, 0, 0, 1], [3.25, -2.5], 7.0, 1, None),                 # pure shift, fractional
    ([1, 0, 0, 1], [-4.0, 6.0], 1.5, 1, None),                  # integer shift
    ([0.99, 0.02, -0.03, 1.01], [1.7, 4.2], 3.0, 1, None),      # small affine (the LinearAlign regime)
    ([0.5, 0, 0, 0.5], [10, 10], 2.0, 1, None),                 # zoom in
    ([1.3, 0.2, -0.1, 1.2], [-5.5, -7.25], 9.0, 1, None),       # zoom out: large fill area
    ([0.0, 1.0, 1.0, 0.0], [0.0, 0.0], 4.0, 1, None),           # transpose
    ([0.99, 0.02, -0.03, 1.01], [1.7, 4.2], 3.0, 0, None),      # nearest-lower mode
    ([1, 0, 0, 1], [-8.5, -6.25], 5.0, 1, (17, 13)),            # output larger than the input (extra margin)
]


def transform_inputs():
    gray = (smooth_noise((97, 131), seed=41, sigma=1.5) * 255.0).astype(np.float32)
    rgb = np.random.default_rng(42).integers(0, 256, (40, 53, 3), dtype=np.uint8)
    return gray, rgb
