"""The pinned result pool (siftmi_host_alloc): bucket sizes, the byte limit with its fall-back to ordinary arrays, the opt-out
(a stack-alignment loop keeps every aligned frame; page-locked memory cannot be swapped)."""
import ctypes as C

import numpy as np
import pytest

from util import smooth_noise, assert_same_keypoints

pytestmark = pytest.mark.gpu


def test_buckets_limit_and_fallback(siftlib):
    import sift_pyocl_amd as sp
    from sift_pyocl_amd import _lib
    L = _lib.lib()
    L.siftmi_host_pool_trim(0)
    live0, spare0 = _lib.pinned_pool()
    assert spare0 == 0
    a = _lib.pinned_empty(3000 * 3000, np.float32)            # 36 000 000 bytes -> 36 MiB, not the next power of two (64 MiB)
    live1, _ = _lib.pinned_pool()
    assert live1 - live0 == 36 << 20
    b = _lib.pinned_empty(1000, sp.SiftPlan.dtype_kp)         # 144 000 bytes -> 256 KiB
    assert _lib.pinned_pool()[0] - live1 == 256 << 10
    del a, b
    live2, spare2 = _lib.pinned_pool()
    assert live2 == live0 and spare2 == (36 << 20) + (256 << 10)       # parked, not handed back from a destructor
    # the limit: nothing more is pinned beyond it, callers get ordinary arrays
    img = smooth_noise((600, 700), seed=8)
    ref = sp.LinearAlign(img)
    _lib.pinned_pool(limit=live0 + (1 << 20))                 # 1 MiB of room: a 1.68 MB frame does not fit
    try:
        with pytest.raises(MemoryError):
            _lib.pinned_empty(600 * 700, np.float32)
        held = _lib.pinned_pool()[0]
        out = ref.transform(np.identity(2, np.float32), np.zeros(2, np.float32), image=img, fill=0.0)
        assert out.shape == img.shape and _lib.pinned_pool()[0] == held          # an ordinary array: nothing more is pinned
        want = sp.SiftPlan(template=img).keypoints(img)       # small results still fit; large ones fall back silently
        assert len(want) > 100
    finally:
        _lib.pinned_pool(limit=2 << 30)
    # the opt-out
    ref.sift.pinned_results = False
    before = _lib.pinned_pool()[0]
    out = ref.transform(np.identity(2, np.float32), np.zeros(2, np.float32), image=img, fill=0.0)
    assert _lib.pinned_pool()[0] == before
    plan = sp.SiftPlan(template=img)
    plan.pinned_results = False
    assert_same_keypoints(plan.keypoints(img), want, "plain result arrays")
    assert _lib.pinned_pool()[0] == before
    L.siftmi_host_pool_trim(0)
    assert _lib.pinned_pool()[1] == 0
