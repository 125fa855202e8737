import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import sift_pyocl_amd as sp
from sift_pyocl_amd import _lib
S = 4096
dev = [torch.from_numpy(np.random.default_rng(i).random((S, S), dtype=np.float32)).cuda() for i in range(8)]
frames = [dev[i % 8] for i in range(16)]
bp = sp.BatchPlan(shape=(S, S), dtype=np.float32, octave_max=3, lanes=2)
L = _lib.lib()
n = len(frames)
ptrs = (C.c_void_p * n)(*[f.data_ptr() for f in frames])
counts = (C.c_int64 * n)(); offsets = (C.c_int64 * n)(); total = C.c_int64(); ovf = C.c_int32()
for rep in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    L.siftmi_batch_keypoints(bp._handle, ptrs, n, 0, 1, counts, offsets, C.byref(total), C.byref(ovf))
    t1 = time.perf_counter()
    flat = np.empty(total.value, dtype=bp.dtype_kp)
    t2 = time.perf_counter()
    L.siftmi_batch_fetch(bp._handle, flat.ctypes.data, 0, 0, total.value)
    t3 = time.perf_counter()
    print("rep %d: batch_keypoints %.2f ms (%.3f/frame)  alloc %.2f ms  fetch %d records %.2f ms" % (rep, 1e3 * (t1 - t0), 1e3 * (t1 - t0) / n, 1e3 * (t2 - t1), total.value, 1e3 * (t3 - t2)))
