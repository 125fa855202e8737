"""ctypes binding of libsiftmi.so (the C ABI declared in include/siftmi.h).

There is no CPU fallback: if the HIP library is missing or no GPU is visible, creating a plan
raises.  ``build()`` compiles the library in-tree with hipcc (cross-compiles without a GPU).
"""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libsiftmi.so")

OK, EINVAL, ENOMEM, EDEVICE, ECAPACITY = 0, -1, -2, -3, -4

DTYPE_CODES = {"float32": 0, "uint8": 1, "uint16": 2, "uint32": 3, "uint64": 4, "int32": 5, "int64": 6,
               "float64": 7, "rgb8": 8}


class Params(C.Structure):
    _fields_ = [("init_sigma", C.c_double), ("peak_thresh", C.c_float), ("edge_thresh0", C.c_float),
                ("edge_thresh", C.c_float), ("ori_sigma", C.c_float), ("border_dist", C.c_int32),
                ("octave_max", C.c_int32), ("pix_per_kp", C.c_int32), ("double_im_size", C.c_int32)]


def build(force=False):
    """Compile sift_pyocl_amd/libsiftmi.so for gfx950."""
    src = os.path.join(HERE, "csrc")
    if force and os.path.exists(LIB_PATH):
        os.remove(LIB_PATH)
    subprocess.check_call(["make", "-s", "-C", src])
    return LIB_PATH


_lib = None

_SIGNATURES = {
    "siftmi_device_count": (C.c_int, []),
    "siftmi_device_name": (C.c_int, [C.c_int, C.c_char_p, C.c_int64]),
    "siftmi_last_error": (C.c_char_p, []),
    "siftmi_version": (C.c_char_p, []),
    "siftmi_plan_create": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(Params), C.c_int32,
                                     C.POINTER(C.c_void_p)]),
    "siftmi_plan_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "siftmi_plan_set_params": (C.c_int, [C.c_void_p, C.POINTER(Params)]),
    "siftmi_plan_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int64]),
    "siftmi_plan_capacity": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "siftmi_plan_tail_timeouts": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int32)]),
    "siftmi_batch_tail_timeouts": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int32)]),
    "siftmi_host_pool_limit": (C.c_int, [C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "siftmi_host_pool_trim": (C.c_int, [C.c_int64]),
    "siftmi_host_alloc": (C.c_int, [C.c_int64, C.POINTER(C.c_void_p)]),
    "siftmi_host_free": (C.c_int, [C.c_void_p]),
    "siftmi_plan_keypoints": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_int64,
                                        C.POINTER(C.c_int64), C.POINTER(C.c_int32)]),
    "siftmi_plan_fetch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_int64]),
    "siftmi_plan_get_minmax": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "siftmi_plan_profile": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int64]),
    "siftmi_plan_last_kernel_ms": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float),
                                             C.POINTER(C.c_int32), C.POINTER(C.c_double)]),
    "siftmi_plan_blur_ms": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_float), C.POINTER(C.c_int32),
                                      C.POINTER(C.c_double)]),
    "siftmi_plan_profile_totals": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                             C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
    "siftmi_plan_destroy": (C.c_int, [C.c_void_p]),
    "siftmi_match_create": (C.c_int, [C.c_int64, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]),
    "siftmi_match": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_int64, C.c_int32,
                               C.c_float, C.c_void_p, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "siftmi_batch_create": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.POINTER(C.c_void_p)]),
    "siftmi_batch_destroy": (C.c_int, [C.c_void_p]),
    "siftmi_batch_set_params": (C.c_int, [C.c_void_p, C.c_void_p]),
    "siftmi_batch_set_profile": (C.c_int, [C.c_void_p, C.c_int32]),
    "siftmi_batch_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int64]),
    "siftmi_batch_blur_ms": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
    "siftmi_batch_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int64)]),
    "siftmi_batch_keypoints": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int64),
                                         C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int32)]),
    "siftmi_batch_keypoints_into": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_void_p),
                                              C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                              C.POINTER(C.c_int32)]),
    "siftmi_batch_fetch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_int64]),
    "siftmi_match_set_roi": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]),
    "siftmi_match_ex": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_int64, C.c_int32, C.c_float, C.c_int32,
                                  C.c_int32, C.c_void_p, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "siftmi_match_last_kernel_ms": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "siftmi_match_last_stage_ms": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "siftmi_match_destroy": (C.c_int, [C.c_void_p]),
    "siftmi_stage_gaussian_taps": (C.c_int, [C.c_float, C.c_int32, C.c_void_p]),
    "siftmi_stage_xcd_order": (C.c_int32, [C.c_int32, C.c_int32]),
    "siftmi_stage_minmax_normalize": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                                C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "siftmi_stage_blur": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32]),
    "siftmi_stage_blur_ex": (C.c_int, [C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32,
                                       C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32)]),
    "siftmi_stage_dog": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]),
    "siftmi_stage_local_maxmin": (C.c_int, [C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(Params),
                                            C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]),
    "siftmi_stage_interp": (C.c_int, [C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int64,
                                      C.POINTER(Params), C.c_void_p, C.c_void_p, C.POINTER(C.c_int64)]),
    "siftmi_stage_compact": (C.c_int, [C.c_int32, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.POINTER(C.c_int64)]),
    "siftmi_stage_gradient": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]),
    "siftmi_stage_orientation": (C.c_int, [C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                                           C.c_void_p, C.c_int64, C.POINTER(Params), C.c_void_p, C.c_void_p,
                                           C.c_int64, C.POINTER(C.c_int64)]),
    "siftmi_stage_descriptor": (C.c_int, [C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                                          C.c_void_p, C.c_int64, C.c_void_p]),
    "siftmi_plan_records_device": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]),
    "siftmi_plan_transform": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                        C.c_void_p, C.c_void_p, C.c_float, C.c_int32, C.POINTER(C.c_double)]),
    "siftmi_stage_shrink": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]),
    "siftmi_stage_convert": (C.c_int, [C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32]),
    "siftmi_stage_math": (C.c_int, [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]),
}


def source_fingerprint():
    """sha256 (16 hex digits) over the sources libsiftmi.so is built from (csrc/*, include/siftmi.h).  Profile summaries that
    bench.py replays beside a live measurement (HBM traffic, VALU instruction counts) carry the fingerprint of the
    library they were taken from; a summary of other kernels is not replayed."""
    import glob
    import hashlib
    here = os.path.dirname(os.path.abspath(__file__))
    files = sorted(glob.glob(os.path.join(here, "csrc", "*.hpp")) + glob.glob(os.path.join(here, "csrc", "*.hip")) +
                   [os.path.join(os.path.dirname(here), "include", "siftmi.h")])
    h = hashlib.sha256()
    for f in files:
        if not os.path.exists(f):        # an installed package without the repository's include/: no fingerprint, nothing replayed
            return None
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def exported_symbols():
    """Every entry point include/siftmi.h declares (checked against the .so by the CPU tests)."""
    return sorted(_SIGNATURES)


def _share_hip_runtime_with_torch():
    """PyTorch-ROCm wheels bundle their own libamdhip64 / libhsa-runtime64, and two HIP runtimes in one
    process cannot both own the GPU (the second one reports "no HIP GPUs").  libsiftmi.so only names
    ``libamdhip64.so.7``: if torch is importable, load it first so the dynamic loader binds our library
    to the runtime torch already mapped; without torch the ROCm install in RUNPATH is used.
    Set SIFTMI_STANDALONE=1 to skip the import."""
    if os.environ.get("SIFTMI_STANDALONE") == "1":
        return
    try:
        import torch  # noqa: F401
    except ImportError:
        pass


def lib():
    """Load libsiftmi.so; raise ImportError loudly when it has not been built."""
    global _lib
    if _lib is None:
        _share_hip_runtime_with_torch()
        if not os.path.exists(LIB_PATH):
            raise ImportError("sift_pyocl_amd: %s is missing -- run `python -c 'import __graft_entry__ as g; "
                              "g.build()'` (or make -C sift_pyocl_amd/csrc); there is no CPU fallback" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


class PinnedBlock(object):
    """A pinned host block from the library's pool, exposed to numpy through ``__array_interface__``; arrays created
    with ``numpy.asarray(block)`` (and their views) keep it alive, the block returns to the pool when the last one dies."""

    def __init__(self, nbytes):
        ptr = C.c_void_p()
        check(lib().siftmi_host_alloc(int(nbytes), C.byref(ptr)))
        self.ptr = ptr.value
        self.nbytes = int(nbytes)
        self.__array_interface__ = {"shape": (self.nbytes,), "typestr": "|u1", "data": (self.ptr, False), "version": 3}

    def __del__(self):
        ptr, self.ptr = getattr(self, "ptr", None), None
        if ptr and _lib is not None:
            try:
                _lib.siftmi_host_free(ptr)
            except Exception:
                pass


def pinned_empty(count, dtype):
    """numpy array of `count` records of `dtype` in pinned, device-writable host memory"""
    import numpy
    dtype = numpy.dtype(dtype)
    block = PinnedBlock(max(1, count) * dtype.itemsize)
    return numpy.asarray(block)[:count * dtype.itemsize].view(dtype)


def pinned_pool(limit=None):
    """(live bytes, spare bytes) of the pinned result pool; `limit` (bytes) caps what it may hold -- beyond it results come
    back as ordinary numpy arrays (one copy after the last kernel)."""
    live, spare = C.c_int64(), C.c_int64()
    check(lib().siftmi_host_pool_limit(-1 if limit is None else int(limit), C.byref(live), C.byref(spare)))
    return int(live.value), int(spare.value)


def last_error():
    msg = lib().siftmi_last_error()
    return msg.decode("utf-8", "replace") if msg else ""


def check(rc, allow=()):
    """Map C-ABI status codes onto the exception types the reference raises."""
    if rc == OK or rc in allow:
        return rc
    msg = last_error()
    if rc == ENOMEM:
        raise MemoryError(msg)
    raise RuntimeError(msg or "libsiftmi error %d" % rc)
