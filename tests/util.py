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
