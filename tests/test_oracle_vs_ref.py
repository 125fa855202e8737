"""CPU test: the oracle against the reference's own kernels run natively (oracle/_ref), on inputs that
are NOT in the golden set.  Skipped where the native reference build is unavailable (it can only be
built where /root/reference is mounted; the prebuilt .so may travel with the repository snapshot)."""
import numpy as np
import pytest

from util import compare_keypoints_libm, smooth_noise, sort_rows, white_noise


@pytest.fixture(scope="module")
def ref():
    from oracle import pyref
    if not pyref.available() and not pyref.build():
        pytest.skip("oracle/_ref/libsiftclref.so not built (needs /root/reference)")
    return pyref


@pytest.mark.parametrize("seed,shape,smooth", [(21, (200, 333), True), (22, (160, 160), False), (23, (97, 211), True)])
def test_pipeline_identical(oracle, ref, seed, shape, smooth):
    img = smooth_noise(shape, seed=seed, sigma=2.5) if smooth else white_noise(shape, seed=seed)
    print(compare_keypoints_libm(oracle.keypoints(img), ref.keypoints(img), "seed %d" % seed))


def test_taps_identical_for_other_sigmas(oracle, ref):
    for sigma in (0.8, 1.0, 2.2, 3.3, 5.0):
        size = ref.kernel_size(sigma)
        assert np.array_equal(oracle.gaussian_taps(sigma, size), ref.gaussian_taps(sigma, size))


def test_match_identical(oracle, ref):
    img = smooth_noise((220, 220), seed=31)
    k1 = oracle.keypoints(img)
    k2 = oracle.keypoints(np.roll(img, (3, 4), axis=(0, 1)))
    p_o, n_o = oracle.match(k1, k2)
    p_r, n_r = ref.match(k1, k2, cap=len(k1))
    assert n_o == n_r and np.array_equal(sort_rows(p_o), sort_rows(p_r))


def test_transform_identical(oracle, ref):
    rng = np.random.default_rng(77)
    img = (smooth_noise((150, 203), seed=78) * 1000.0 - 300.0).astype(np.float32)
    rgb = rng.integers(0, 256, (64, 81, 3), dtype=np.uint8)
    for k in range(12):
        M = (np.eye(2) + rng.normal(0, 0.08, (2, 2))).astype(np.float32).reshape(4)
        off = rng.normal(0, 12, 2).astype(np.float32)
        fill = float(rng.integers(0, 200))
        for mode in (1, 0):
            a = oracle.transform(img, M, off, fill=fill, mode=mode); b = ref.transform(img, M, off, fill=fill, mode=mode)
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (k, mode)
            assert np.array_equal(oracle.transform(rgb, M, off, fill=fill, mode=mode), ref.transform(rgb, M, off, fill=fill, mode=mode))
        a = oracle.transform(img, M, off, out_shape=(180, 230), fill=fill); b = ref.transform(img, M, off, out_shape=(180, 230), fill=fill)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
