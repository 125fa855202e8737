"""ctypes access to the TEST-ONLY CPU oracle (oracle/libsiftoracle.so).

Test infrastructure: may be imported only by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package (sift_pyocl_amd) never imports this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libsiftoracle.so")

dtype_kp = np.dtype([("x", np.float32), ("y", np.float32), ("scale", np.float32),
                     ("angle", np.float32), ("desc", (np.uint8, 128))])
dtype_kp4 = np.dtype((np.float32, 4))


class Params(C.Structure):
    _fields_ = [("init_sigma", C.c_double), ("peak_thresh", C.c_float), ("edge_thresh0", C.c_float),
                ("edge_thresh", C.c_float), ("ori_sigma", C.c_float), ("border_dist", C.c_int),
                ("octave_max", C.c_int), ("pix_per_kp", C.c_int), ("double_im_size", C.c_int)]


def default_params(octave_max=0, pix_per_kp=10, init_sigma=1.6, double_im_size=0):
    """Values of sift-src/param.py:52-79 as plan.py passes them to the kernels."""
    return Params(init_sigma=float(init_sigma), peak_thresh=np.float32(255.0 * 0.04 / 3.0),
                  edge_thresh0=np.float32(0.08), edge_thresh=np.float32(0.06), ori_sigma=np.float32(1.5),
                  border_dist=5, octave_max=int(octave_max), pix_per_kp=int(pix_per_kp), double_im_size=int(bool(double_im_size)))


def build(force=False):
    if force or not os.path.exists(LIB_PATH):
        subprocess.check_call(["make", "-s", "-C", HERE, "oracle"])
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        fp = C.POINTER(C.c_float)
        L.so_expf.restype = C.c_float; L.so_expf.argtypes = [C.c_float]
        L.so_exp2f.restype = C.c_float; L.so_exp2f.argtypes = [C.c_float]
        L.so_atan2f.restype = C.c_float; L.so_atan2f.argtypes = [C.c_float, C.c_float]
        L.so_sincosf.restype = None; L.so_sincosf.argtypes = [C.c_float, fp, fp]
        L.so_match.restype = C.c_int64
        L.so_kernel_size.argtypes = [C.c_double]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


# ----------------------------------------------------------------------------- stages
def expf_array(x):
    x = np.ascontiguousarray(x, np.float32); out = np.empty_like(x)
    lib().so_expf_array(_p(x), _p(out), C.c_int64(x.size))
    return out


def atan2f_array(y, x):
    y = np.ascontiguousarray(y, np.float32); x = np.ascontiguousarray(x, np.float32); out = np.empty_like(x)
    lib().so_atan2f_array(_p(y), _p(x), _p(out), C.c_int64(x.size))
    return out


def gaussian_taps(sigma, size):
    out = np.empty(size, np.float32)
    lib().so_gaussian_taps(C.c_float(np.float32(sigma)), C.c_int(size), _p(out))
    return out


def minmax(img):
    img = _f32(img)
    mn, mx = C.c_float(), C.c_float()
    lib().so_minmax(_p(img), C.c_int64(img.size), C.byref(mn), C.byref(mx))
    return np.float32(mn.value), np.float32(mx.value)


def normalize(img, mn, mx):
    out = _f32(img).copy()
    lib().so_normalize(_p(out), C.c_int64(out.size), C.c_float(mn), C.c_float(mx))
    return out


def blur(img, taps):
    img = _f32(img); taps = _f32(taps)
    H, W = img.shape
    out = np.empty_like(img); tmp = np.empty_like(img)
    lib().so_blur(_p(img), _p(out), _p(tmp), _p(taps), C.c_int(taps.size), C.c_int(W), C.c_int(H))
    return out


def convolve_h(img, taps):
    img = _f32(img); taps = _f32(taps)
    H, W = img.shape
    out = np.empty_like(img)
    lib().so_convolve_h(_p(img), _p(out), _p(taps), C.c_int(taps.size), C.c_int(W), C.c_int(H))
    return out


def convolve_v(img, taps):
    img = _f32(img); taps = _f32(taps)
    H, W = img.shape
    out = np.empty_like(img)
    lib().so_convolve_v(_p(img), _p(out), _p(taps), C.c_int(taps.size), C.c_int(W), C.c_int(H))
    return out


def dog(blurs):
    """blurs: (6,H,W) -> (5,H,W) with DoG[s] = blur[s] - blur[s+1] via combine()."""
    blurs = _f32(blurs)
    out = np.empty((5,) + blurs.shape[1:], np.float32)
    n = blurs[0].size
    for s in range(5):
        lib().so_combine(_p(blurs[s + 1]), C.c_float(-1.0), _p(blurs[s]), C.c_float(1.0), _p(out[s]), C.c_int64(n))
    return out


def local_maxmin(dogs, scale, octsize, capacity, par=None, kps=None, counter=0):
    par = par or default_params()
    dogs = _f32(dogs)
    _, H, W = dogs.shape
    if kps is None:
        kps = np.full((capacity, 4), -1, np.float32)
    cnt = C.c_int(counter)
    lib().so_local_maxmin(_p(dogs), _p(kps), C.c_int(par.border_dist), C.c_float(par.peak_thresh), C.c_int(octsize),
                          C.c_float(par.edge_thresh0), C.c_float(par.edge_thresh), C.byref(cnt), C.c_int(capacity),
                          C.c_int(scale), C.c_int(W), C.c_int(H))
    return kps, cnt.value


def interp_keypoint(dogs, kps, start, end, par=None):
    par = par or default_params()
    dogs = _f32(dogs)
    _, H, W = dogs.shape
    kps = _f32(kps).copy()
    lib().so_interp_keypoint(_p(dogs), _p(kps), C.c_int(start), C.c_int(end), C.c_float(par.peak_thresh),
                             C.c_float(np.float32(par.init_sigma)), C.c_int(W), C.c_int(H))
    return kps


def compact(kps, start, end):
    kps = _f32(kps)
    out = np.full_like(kps, -1)
    n = lib().so_compact(_p(kps), _p(out), C.c_int(start), C.c_int(end))
    return out, n


def gradient(img):
    img = _f32(img)
    H, W = img.shape
    g = np.empty_like(img); o = np.empty_like(img)
    lib().so_gradient(_p(img), _p(g), _p(o), C.c_int(W), C.c_int(H))
    return g, o


def orientation(kps, grad, ori, octsize, start, end, capacity=None, par=None):
    par = par or default_params()
    kps = _f32(kps).copy()
    grad = _f32(grad); ori = _f32(ori)
    H, W = grad.shape
    capacity = capacity or kps.shape[0]
    cnt = C.c_int(end)
    lib().so_orientation(_p(kps), _p(grad), _p(ori), C.byref(cnt), C.c_int(octsize), C.c_float(par.ori_sigma),
                         C.c_int(capacity), C.c_int(start), C.c_int(end), C.c_int(W), C.c_int(H))
    return kps, cnt.value


def descriptor(kps, grad, ori, octsize, start, end):
    kps = _f32(kps)
    grad = _f32(grad); ori = _f32(ori)
    H, W = grad.shape
    desc = np.zeros((kps.shape[0], 128), np.uint8)
    lib().so_descriptor(_p(kps), _p(desc), _p(grad), _p(ori), C.c_int(octsize), C.c_int(start), C.c_int(end),
                        C.c_int(W), C.c_int(H))
    return desc


def shrink(img):
    img = _f32(img)
    H, W = img.shape
    out = np.empty((H // 2, W // 2), np.float32)
    lib().so_shrink(_p(img), _p(out), C.c_int(W), C.c_int(H), C.c_int(W // 2), C.c_int(H // 2))
    return out


# ----------------------------------------------------------------------------- pipeline
def keypoints(image, par=None, return_overflow=False):
    par = par or default_params()
    image = _f32(image)
    H, W = image.shape
    n_oct = lib().so_octave_count(C.c_int(H), C.c_int(W))
    if par.octave_max > 0:
        n_oct = min(n_oct, par.octave_max)
    cap = max(1, (H * W // par.pix_per_kp)) * max(1, n_oct)
    out = np.empty(cap, dtype_kp)
    n = C.c_int64(0); ovf = C.c_int(0)
    rc = lib().so_keypoints(_p(image), C.c_int(H), C.c_int(W), C.byref(par), _p(out), C.c_int64(cap),
                            C.byref(n), C.byref(ovf))
    if rc != 0:
        raise MemoryError("oracle allocation failed")
    res = out[:n.value].copy().view(np.recarray)
    return (res, bool(ovf.value)) if return_overflow else res


def match(kp1, kp2, ratio_th=np.float32(0.73 * 0.73), cap=None):
    kp1 = np.ascontiguousarray(kp1, dtype=dtype_kp); kp2 = np.ascontiguousarray(kp2, dtype=dtype_kp)
    cap = cap if cap is not None else max(1, kp1.size)
    pairs = np.full((cap, 2), -1, np.int32)
    n = lib().so_match(_p(kp1), C.c_int64(kp1.size), _p(kp2), C.c_int64(kp2.size), C.c_float(ratio_th),
                       _p(pairs), C.c_int64(cap))
    return pairs[:min(n, cap)].copy(), int(n)


def match_ex(kp1, kp2, roi=None, roi_mode=0, mutual=False, ratio_th=np.float32(0.73 * 0.73), cap=None):
    """ROI-masked (roi_mode 1: matching_valid literal, 2: strict) and / or mutual-best matching."""
    kp1 = np.ascontiguousarray(kp1, dtype=dtype_kp); kp2 = np.ascontiguousarray(kp2, dtype=dtype_kp)
    cap = cap if cap is not None else max(1, kp1.size)
    pairs = np.full((cap, 2), -1, np.int32)
    if roi is not None:
        roi = np.ascontiguousarray(roi, np.int8)
        rh, rw = roi.shape
    else:
        rh = rw = 0
    lib().so_match_ex.restype = C.c_int64
    n = lib().so_match_ex(_p(kp1), C.c_int64(kp1.size), _p(kp2), C.c_int64(kp2.size), C.c_float(ratio_th),
                          _p(roi) if roi is not None else None, C.c_int(rw), C.c_int(rh), C.c_int(roi_mode if roi is not None else 0),
                          C.c_int(int(mutual)), _p(pairs), C.c_int64(cap))
    return pairs[:min(n, cap)].copy(), int(n)


def octave_count(H, W):
    return lib().so_octave_count(C.c_int(H), C.c_int(W))


def transform(image, matrix, offset, out_shape=None, fill=0.0, mode=1):
    """transform.cl restated (float32 image, or uint8 RGB)."""
    matrix = _f32(matrix).reshape(4); offset = _f32(offset).reshape(2)
    if image.ndim == 3:
        image = np.ascontiguousarray(image, np.uint8)
        H, W = image.shape[:2]
        OH, OW = out_shape or (H, W)
        out = np.empty((OH, OW, 3), np.uint8)
        lib().so_transform_rgb(_p(image), _p(out), _p(matrix), _p(offset), C.c_int(W), C.c_int(H), C.c_int(OW), C.c_int(OH),
                               C.c_float(fill), C.c_int(mode))
        return out
    image = _f32(image)
    H, W = image.shape
    OH, OW = out_shape or (H, W)
    out = np.empty((OH, OW), np.float32)
    lib().so_transform(_p(image), _p(out), _p(matrix), _p(offset), C.c_int(W), C.c_int(H), C.c_int(OW), C.c_int(OH),
                       C.c_float(fill), C.c_int(mode))
    return out
