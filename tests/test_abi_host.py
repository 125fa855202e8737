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
    from sift_pyocl_amd import _lib
    h = C.c_void_p()
    par = _lib.Params(init_sigma=1.6, peak_thresh=3.4, edge_thresh0=0.08, edge_thresh=0.06, ori_sigma=1.5,
                      border_dist=5, octave_max=0, pix_per_kp=10, reserved=0)
    assert L.siftmi_plan_create(64, 64, 0, 0, C.byref(par), 0, C.byref(h)) == _lib.EDEVICE
    assert b"device" in L.siftmi_last_error().lower()
    out = np.empty((4, 4), np.float32)
    assert L.siftmi_stage_blur(0, out.ctypes.data, out.ctypes.data, 4, 4, out.ctypes.data, 3) != 0


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
