"""dev: kernel time of the marching blur vs plane height (run under rocprofv3 --kernel-trace): separates the per-launch
fixed cost from the per-row cost."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from sift_pyocl_amd import _lib
L = _lib.lib()
rng = np.random.default_rng(0)
for ntaps, sigma in ((11, 1.2262735), (27, 3.0900156)):
    taps = np.empty(ntaps, np.float32)
    assert L.siftmi_stage_gaussian_taps(C.c_float(sigma), ntaps, taps.ctypes.data) == 0
    for H in (1024, 2048, 4096, 8192, 16384):
        W = 4096
        img = rng.random((H, W), dtype=np.float32)
        out = np.empty_like(img)
        for _ in range(2):
            assert L.siftmi_stage_blur(0, img.ctypes.data, out.ctypes.data, W, H, taps.ctypes.data, ntaps) == 0
