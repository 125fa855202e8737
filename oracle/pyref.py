"""Driver for the natively compiled reference kernels (oracle/_ref/libsiftclref.so).

TEST INFRASTRUCTURE.  The library holds the reference's own OpenCL C kernels built for
x86-64 (oracle/Makefile `ref` target); this module sequences them in the launch order and
with the scalar arguments of the reference host code (sift-src/plan.py:432-756,
sift-src/match.py:200-271), replacing PyOpenCL by direct calls.  It is used to pin the CPU
oracle (tests/test_oracle_vs_ref.py) and to generate tests/golden/*.npz.

It can only be (re)built where /root/reference is mounted; the built .so travels to the GPU
box but nothing at run time reads /root/reference.
"""
import ctypes as C
import math
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "_ref", "libsiftclref.so")

dtype_kp = np.dtype([("x", np.float32), ("y", np.float32), ("scale", np.float32),
                     ("angle", np.float32), ("desc", (np.uint8, 128))])

# sift-src/param.py:52-79
PAR = dict(InitSigma=1.6, BorderDist=5, Scales=3, PeakThresh=255.0 * 0.04 / 3.0, EdgeThresh=0.06,
           EdgeThresh1=0.08, OriSigma=1.5, MatchRatio=0.73, DoubleImSize=0)


# Two builds of the same reference objects (oracle/Makefile): "glibc" binds the OpenCL math builtins to libm,
# "siftmath" binds exp/sin/cos/atan2/pow(2,.) to the oracle's siftmath functions (libm isolated).
LIB_PATHS = {"glibc": LIB_PATH, "siftmath": os.path.join(HERE, "_ref", "libsiftclref_sm.so")}
_variant = "glibc"
_libs = {}


def available(variant=None):
    return os.path.exists(LIB_PATHS[variant or _variant])


def build():
    if os.path.isdir("/root/reference/openCL"):
        subprocess.check_call(["make", "-s", "-C", HERE, "ref"])
    return available()


def use(variant):
    """Select the build every function of this module drives: "glibc" (default) or "siftmath"."""
    global _variant
    assert variant in LIB_PATHS
    _variant = variant


def lib():
    if _variant not in _libs:
        if not available() and not build():
            raise RuntimeError("%s is not built (needs /root/reference)" % LIB_PATHS[_variant])
        _libs[_variant] = C.CDLL(LIB_PATHS[_variant])
    return _libs[_variant]


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def nextpower(n):  # utils.py:69-75
    return 1 << int(math.ceil(math.log(n, 2)))


def kernel_size(sigma, odd=True, cutoff=4):  # utils.py:54-64
    size = int(math.ceil(2 * cutoff * sigma + 1))
    if odd and size % 2 == 0:
        size += 1
    return size


def gaussian_taps(sigma, size=None):
    size = size or kernel_size(sigma)
    out = np.zeros(size, np.float32)
    rc = lib().ref_gaussian(_p(out), C.c_float(np.float32(sigma)), C.c_int(size), C.c_int(nextpower(size)))
    assert rc == 0
    return out


def sigma_schedule(init_sigma=None):
    """(init blur sigma or None, [5 per-octave sigmas]) as plan.py:534-539 / 602-618."""
    init_sigma = float(PAR["InitSigma"] if init_sigma is None else init_sigma)
    ratio = 2.0 ** (1.0 / PAR["Scales"])
    cur = 1.0 if PAR["DoubleImSize"] else 0.5
    first = math.sqrt(init_sigma ** 2 - cur ** 2) if init_sigma > cur else None
    sig, prev = [], init_sigma
    for _ in range(PAR["Scales"] + 2):
        sig.append(prev * math.sqrt(ratio ** 2 - 1.0))
        prev *= ratio
    return first, sig


def minmax(img):
    mx = np.zeros(1, np.float32); mn = np.zeros(1, np.float32)
    lib().ref_max_min_serial(_p(img), C.c_uint(img.size), _p(mx), _p(mn))
    return mn[0], mx[0]


def normalize(img, mn, mx):
    out = np.ascontiguousarray(img, np.float32).copy()
    H, W = out.shape
    a = np.array([mn], np.float32); b = np.array([mx], np.float32); t = np.array([255.0], np.float32)
    lib().ref_normalizes(_p(out), _p(a), _p(b), _p(t), C.c_int(W), C.c_int(H))
    return out


def convolve_h(img, taps):
    img = np.ascontiguousarray(img, np.float32); taps = np.ascontiguousarray(taps, np.float32)
    H, W = img.shape
    out = np.empty_like(img)
    lib().ref_horizontal_convolution(_p(img), _p(out), _p(taps), C.c_int(taps.size), C.c_int(W), C.c_int(H))
    return out


def convolve_v(img, taps):
    img = np.ascontiguousarray(img, np.float32); taps = np.ascontiguousarray(taps, np.float32)
    H, W = img.shape
    out = np.empty_like(img)
    lib().ref_vertical_convolution(_p(img), _p(out), _p(taps), C.c_int(taps.size), C.c_int(W), C.c_int(H))
    return out


def blur(img, taps):
    return convolve_v(convolve_h(img, taps), taps)


def gradient(img):
    img = np.ascontiguousarray(img, np.float32)
    H, W = img.shape
    g = np.empty_like(img); o = np.empty_like(img)
    lib().ref_compute_gradient_orientation(_p(img), _p(g), _p(o), C.c_int(W), C.c_int(H))
    return g, o


def shrink(img):
    img = np.ascontiguousarray(img, np.float32)
    H, W = img.shape
    out = np.empty((H // 2, W // 2), np.float32)
    lib().ref_shrink(_p(img), _p(out), C.c_int(W), C.c_int(H), C.c_int(W // 2), C.c_int(H // 2))
    return out


_CONVERTERS = {"uint8": "ref_u8_to_float", "uint16": "ref_u16_to_float", "uint32": "ref_u32_to_float",
               "uint64": "ref_u64_to_float", "int32": "ref_s32_to_float", "int64": "ref_s64_to_float"}


def to_float(img):
    """Integer / RGB frame -> float32 through the reference's converter kernels (preprocess.cl:53-223),
    as plan.py:464-486 launches them."""
    img = np.ascontiguousarray(img)
    H, W = img.shape[:2]
    out = np.empty((H, W), np.float32)
    if img.ndim == 3:
        assert img.dtype == np.uint8 and img.shape[2] == 3
        lib().ref_rgb_to_float(_p(img), _p(out), C.c_int(W), C.c_int(H))
    else:
        getattr(lib(), _CONVERTERS[img.dtype.name])(_p(img), _p(out), C.c_int(W), C.c_int(H))
    return out


def octave_shapes(shape):
    """plan.py:213-224 (shapes kept in (H, W) order here)."""
    shapes = [tuple(int(i) for i in shape)]
    s = shapes[0]
    while min(s) > 2 * PAR["BorderDist"] + 2:
        s = tuple(i // 2 for i in s)
        shapes.append(s)
    shapes.pop()
    return shapes


def keypoints(image, octave_max=0, pix_per_kp=10, init_sigma=None, stages=None):
    """Full pipeline through the reference kernels.  `stages`, if a dict, receives the
    intermediates of each octave (taps, blurs, DoGs, candidate / refined / oriented lists)."""
    L = lib()
    img = np.ascontiguousarray(image, np.float32)
    H0, W0 = img.shape
    shapes = octave_shapes(img.shape)
    if octave_max:
        shapes = shapes[:octave_max]
    kpsize = int(H0 * W0 // pix_per_kp)
    init_sigma_d = float(PAR["InitSigma"] if init_sigma is None else init_sigma)
    first, sigmas = sigma_schedule(init_sigma_d)
    taps = [gaussian_taps(s) for s in sigmas]
    mn, mx = minmax(img)
    base = normalize(img, mn, mx)
    if first is not None:
        t0 = gaussian_taps(first)
        base = blur(base, t0)
    if stages is not None:
        stages["min"], stages["max"] = mn, mx
        stages["taps"] = taps
        stages["taps_init"] = t0 if first is not None else None
        stages["base"] = base.copy()
        stages["octaves"] = []
    peak = np.float32(PAR["PeakThresh"])
    results = []
    octsize = 1
    for octave, (H, W) in enumerate(shapes):
        n = H * W
        blurs = np.empty((6, H, W), np.float32)
        blurs[0] = base
        dogs = np.empty((5, H, W), np.float32)
        kp1 = np.full((kpsize, 4), -1, np.float32)
        kp2 = np.full((kpsize, 4), -1, np.float32)
        desc = np.zeros((kpsize, 128), np.uint8)
        cnt = np.zeros(1, np.int32)
        for s in range(5):
            blurs[s + 1] = blur(blurs[s], taps[s])
            L.ref_combine(_p(blurs[s + 1]), C.c_float(-1.0), _p(blurs[s]), C.c_float(1.0), _p(dogs), C.c_int(s),
                          C.c_int(W), C.c_int(H))
        st = dict(shape=(H, W), blurs=blurs, dogs=dogs, scales=[]) if stages is not None else None
        last_start = 0
        for s in range(1, 4):
            L.ref_local_maxmin(_p(dogs), _p(kp1), C.c_int(PAR["BorderDist"]), C.c_float(peak), C.c_int(octsize),
                               C.c_float(np.float32(PAR["EdgeThresh1"])), C.c_float(np.float32(PAR["EdgeThresh"])),
                               _p(cnt), C.c_int(kpsize), C.c_int(s), C.c_int(W), C.c_int(H))
            end = int(cnt[0])
            cand = kp1[last_start:end].copy()
            L.ref_interp_keypoint(_p(dogs), _p(kp1), C.c_int(last_start), C.c_int(end), C.c_float(peak),
                                  C.c_float(np.float32(init_sigma_d)), C.c_int(W), C.c_int(H), C.c_int(kpsize))
            interp = kp1[last_start:end].copy()
            cnt[0] = last_start                                     # plan.py:773-774
            L.ref_compact(_p(kp1), _p(kp2), _p(cnt), C.c_int(last_start), C.c_int(end), C.c_int(kpsize))
            newcnt = int(cnt[0])
            kp1, kp2 = kp2, kp1
            kp2[:] = -1
            grad, ori = gradient(blurs[s])
            refined = kp1[last_start:newcnt].copy()
            if newcnt > last_start:
                L.ref_orientation_assignment(_p(kp1), _p(grad), _p(ori), _p(cnt), C.c_int(octsize),
                                             C.c_float(np.float32(PAR["OriSigma"])), C.c_int(kpsize),
                                             C.c_int(last_start), C.c_int(newcnt), C.c_int(W), C.c_int(H),
                                             C.c_int(newcnt))
                L.ref_descriptor(_p(kp1), _p(desc), _p(grad), _p(ori), C.c_int(octsize), C.c_int(last_start),
                                 _p(cnt), C.c_int(W), C.c_int(H), C.c_int(int(cnt[0])))
            if st is not None:
                st["scales"].append(dict(scale=s, candidates=cand, interp=interp, refined=refined,
                                         oriented=kp1[last_start:int(cnt[0])].copy(),
                                         desc=desc[last_start:int(cnt[0])].copy(), grad=grad, ori=ori))
            last_start = int(cnt[0])
        if octave < len(shapes) - 1:
            base = shrink(blurs[3])
        if st is not None:
            stages["octaves"].append(st)
        kp = kp1[:last_start]
        keep = ~np.isnan(kp.sum(axis=-1))                           # plan.py:545-550
        results.append((kp[keep].copy(), desc[:last_start][keep].copy()))
        octsize *= 2
    total = sum(len(k) for k, _ in results)
    out = np.recarray((total,), dtype=dtype_kp)
    at = 0
    for k, d in results:
        m = len(k)
        out.x[at:at + m] = k[:, 0]; out.y[at:at + m] = k[:, 1]
        out.scale[at:at + m] = k[:, 2]; out.angle[at:at + m] = k[:, 3]
        out.desc[at:at + m] = d
        at += m
    return out


def match(kp1, kp2, ratio=None, cap=None):
    kp1 = np.ascontiguousarray(kp1, dtype=dtype_kp); kp2 = np.ascontiguousarray(kp2, dtype=dtype_kp)
    cap = cap or max(1, min(kp1.size, kp2.size))
    ratio = np.float32(PAR["MatchRatio"] * PAR["MatchRatio"]) if ratio is None else np.float32(ratio)
    out = np.full((cap, 2), -1, np.int32)
    cnt = np.zeros(1, np.int32)
    lib().ref_matching(_p(kp1), _p(kp2), _p(out), _p(cnt), C.c_int(cap), C.c_float(ratio), C.c_int(kp1.size),
                       C.c_int(kp2.size), C.c_int(kp1.size))
    n = int(cnt[0])
    return out[:min(n, cap)].copy(), n


def match_valid(kp1, kp2, roi, ratio=None, cap=None):
    """matching_valid (matching_cpu.cl:136-199) run natively; never launched by the reference's host code."""
    kp1 = np.ascontiguousarray(kp1, dtype=dtype_kp); kp2 = np.ascontiguousarray(kp2, dtype=dtype_kp)
    roi = np.ascontiguousarray(roi, np.int8)
    cap = cap or max(1, kp1.size)
    ratio = np.float32(PAR["MatchRatio"] * PAR["MatchRatio"]) if ratio is None else np.float32(ratio)
    out = np.full((cap, 2), -1, np.int32)
    cnt = np.zeros(1, np.int32)
    lib().ref_matching_valid(_p(kp1), _p(kp2), _p(roi), C.c_int(roi.shape[1]), C.c_int(roi.shape[0]), _p(out), _p(cnt), C.c_int(cap),
                             C.c_float(ratio), C.c_int(kp1.size), C.c_int(kp2.size))
    n = int(cnt[0])
    return out[:min(n, cap)].copy(), n


def transform(image, matrix, offset, out_shape=None, fill=0.0, mode=1):
    """transform / transform_RGB of transform.cl run natively (launch as alignment.py:336-346)."""
    matrix = np.ascontiguousarray(matrix, np.float32).reshape(4); offset = np.ascontiguousarray(offset, np.float32).reshape(2)
    if image.ndim == 3:
        image = np.ascontiguousarray(image, np.uint8)
        H, W = image.shape[:2]
        OH, OW = out_shape or (H, W)
        out = np.zeros((OH, OW, 3), np.uint8)
        lib().ref_transform_RGB(_p(image), _p(out), _p(matrix), _p(offset), C.c_int(W), C.c_int(H), C.c_int(OW), C.c_int(OH),
                                C.c_float(fill), C.c_int(mode))
        return out
    image = np.ascontiguousarray(image, np.float32)
    H, W = image.shape
    OH, OW = out_shape or (H, W)
    out = np.zeros((OH, OW), np.float32)
    lib().ref_transform(_p(image), _p(out), _p(matrix), _p(offset), C.c_int(W), C.c_int(H), C.c_int(OW), C.c_int(OH),
                        C.c_float(fill), C.c_int(mode))
    return out
