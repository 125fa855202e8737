"""CPU tests of the product's host side: the C ABI library loads and exports every symbol that
include/siftmi.h declares, fails loudly without a GPU, and its host-only pieces (Gaussian taps,
sizing helpers, record layout) agree with the oracle / the reference's formulas.  No kernels run."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L():
    from sift_pyocl_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    return _lib.lib()


def test_every_declared_symbol_is_exported(L):
    from sift_pyocl_amd import _lib
    header = open(os.path.join(ROOT, "include", "siftmi.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(siftmi_[a-z_0-9]+)\s*\(", header))
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(L, name), "libsiftmi.so does not export %s" % name
    assert declared == set(_lib.exported_symbols()), "ctypes signature table out of sync with include/siftmi.h"


def test_record_and_param_layout():
    from sift_pyocl_amd import SiftPlan, MatchPlan, _lib
    assert SiftPlan.dtype_kp.itemsize == 144 and MatchPlan.dtype_kp == SiftPlan.dtype_kp
    assert SiftPlan.dtype_kp.fields["desc"][1] == 16
    assert C.sizeof(_lib.Params) == 40


def test_no_gpu_fails_loudly(L):
    import sift_pyocl_amd as sp
    if L.siftmi_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(RuntimeError):
        sp.SiftPlan((64, 64), np.float32)
    with pytest.raises(RuntimeError):
        sp.MatchPlan()
    with pytest.raises(RuntimeError):
        sp.BatchPlan(shape=(64, 64), dtype=np.float32, lanes=2)
    with pytest.raises(RuntimeError):
        sp.LinearAlign(np.zeros((64, 64), np.float32))
    from sift_pyocl_amd import _lib
    h = C.c_void_p()
    par = _lib.Params(init_sigma=1.6, peak_thresh=3.4, edge_thresh0=0.08, edge_thresh=0.06, ori_sigma=1.5,
                      border_dist=5, octave_max=0, pix_per_kp=10, double_im_size=0)
    assert L.siftmi_plan_create(64, 64, 0, 0, C.byref(par), 0, C.byref(h)) == _lib.EDEVICE
    assert b"device" in L.siftmi_last_error().lower()
    out = np.empty((4, 4), np.float32)
    assert L.siftmi_stage_blur(0, out.ctypes.data, out.ctypes.data, 4, 4, out.ctypes.data, 3) != 0
    hb = C.c_void_p()
    assert L.siftmi_batch_create(64, 64, 0, 0, C.byref(par), 2, C.byref(hb)) == _lib.EDEVICE and not hb.value
    hm = C.c_void_p()
    assert L.siftmi_match_create(100, 0, 0, C.byref(hm)) == _lib.EDEVICE and not hm.value


def test_matching_correction_is_a_least_squares_fit():
    """utils.matching_correction (completed: the reference's version has no solve, utils.py:156-189) against numpy.linalg.lstsq"""
    from sift_pyocl_amd.utils import affine_least_squares, matching_correction
    from sift_pyocl_amd.match import MatchPlan
    rng = np.random.default_rng(3)
    n = 400
    m = np.zeros((n, 2), MatchPlan.dtype_kp).view(np.recarray)
    x = rng.random(n) * 4000; y = rng.random(n) * 3000
    m.x[:, 0] = x; m.y[:, 0] = y
    m.x[:, 1] = 1.01 * x + 0.02 * y + 3 + rng.normal(0, 0.2, n); m.y[:, 1] = -0.015 * x + 0.99 * y - 4 + rng.normal(0, 0.2, n)
    X = np.zeros((2 * n, 6)); rhs = np.zeros(2 * n)
    X[::2, 0] = m.x[:, 0]; X[::2, 1] = m.y[:, 0]; X[::2, 2] = 1; X[1::2, 3] = m.x[:, 0]; X[1::2, 4] = m.y[:, 0]; X[1::2, 5] = 1
    rhs[::2] = m.x[:, 1]; rhs[1::2] = m.y[:, 1]
    want = np.linalg.lstsq(X, rhs, rcond=None)[0]
    got = matching_correction(m)
    assert np.allclose(got, want, rtol=0, atol=1e-9)
    assert np.allclose(got, [1.01, 0.02, 3, -0.015, 0.99, -4], atol=0.05)
    # degenerate (collinear) input falls back to the minimum-norm solution instead of dividing by zero
    t = affine_least_squares([0, 1, 2, 3], [0, 1, 2, 3], [1, 2, 3, 4], [2, 3, 4, 5])
    assert np.all(np.isfinite(t)) and np.allclose([t[0] * 2 + t[1] * 2 + t[2], t[3] * 2 + t[4] * 2 + t[5]], [3, 4])


def test_host_gaussian_taps_match_oracle(L, oracle):
    """siftmi_stage_gaussian_taps runs on the host (no GPU needed): product taps == oracle taps."""
    import math
    sig = [math.sqrt(1.6 ** 2 - 0.25)]
    prev, ratio = 1.6, 2.0 ** (1.0 / 3.0)
    for _ in range(5):
        sig.append(prev * math.sqrt(ratio ** 2 - 1.0)); prev *= ratio
    for s in sig + [0.7, 2.9, 4.0]:
        size = int(math.ceil(8 * s + 1)); size += (size % 2 == 0)
        out = np.empty(size, np.float32)
        assert L.siftmi_stage_gaussian_taps(C.c_float(s), size, out.ctypes.data) == 0
        assert np.array_equal(out, oracle.gaussian_taps(s, size))
    assert L.siftmi_stage_gaussian_taps(C.c_float(1.0), 0, out.ctypes.data) != 0


def test_xcd_contiguous_order_is_a_bijection(L):
    """csrc/k_xcd.hpp: workgroup id -> tile.  Every grid size must map [0, n) onto itself (a tile worked on twice or never would
    be a wrong plane), XCD c = id % 8 must own one contiguous range, ascending in id."""
    for n in list(range(1, 70)) + [255, 256, 257, 768, 1020, 1024, 2112, 4097]:
        order = [L.siftmi_stage_xcd_order(i, n) for i in range(n)]
        assert sorted(order) == list(range(n)), n
        for c in range(min(8, n)):
            mine = order[c::8]
            assert mine == list(range(mine[0], mine[0] + len(mine))), (n, c)
        starts = [order[c] for c in range(min(8, n))]
        assert starts == sorted(starts), n
    assert L.siftmi_stage_xcd_order(5, 5) == -1 and L.siftmi_stage_xcd_order(-1, 5) == -1 and L.siftmi_stage_xcd_order(0, 0) == -1


def test_utils_follow_reference_formulas():
    from sift_pyocl_amd.utils import calc_size, kernel_size, nextpower
    assert [kernel_size(s, True) for s in (1.5198684, 1.2262735, 1.5450078, 1.9465878, 2.452547, 3.0900156)] == [15, 11, 15, 17, 21, 27]
    assert kernel_size(3.0, False) == 25 and kernel_size(3.0, True) == 25 and kernel_size(1.0, True) == 9
    assert [nextpower(n) for n in (1, 2, 3, 100, 128, 129)] == [1, 2, 4, 128, 128, 256]
    assert calc_size((100, 7), (64, 1)) == (128, 7) and calc_size((100,), 128) == (128,)


def test_param_object():
    from sift_pyocl_amd import par
    assert par.InitSigma == 1.6 and par["BorderDist"] == 5 and par.MatchRatio == 0.73
    assert abs(par.PeakThresh - 255.0 * 0.04 / 3.0) < 1e-15
    with pytest.raises(AttributeError):
        par.NoSuchThing
