"""SiftPlan -- keypoint extraction on one MI355X through libsiftmi.so.

Host-side mirror of the reference's ``sift_pyocl.SiftPlan`` (sift-src/plan.py:70-867): same
constructor keywords, attributes, ``keypoints()`` / ``__call__`` contract, exceptions and output
record type, so code written against the reference runs unchanged.  All arithmetic happens in the
hand-written HIP kernels; this class only validates arguments, marshals pointers and wraps the
result.  There is no PyOpenCL, no CPU fallback and no per-stage host round trip.
"""
import ctypes as C
import logging
import math
import os
import threading
import time

import numpy

from . import _lib
from .param import par
from .utils import calc_size, kernel_size, nextpower

logger = logging.getLogger("sift.plan")


def _pointer_of(image):
    """(pointer, is_device, dtype, shape, keepalive) for numpy arrays, torch tensors (host or
    HIP) or any object exposing __cuda_array_interface__ (the analogue of the reference accepting
    pyopencl.array.Array inputs, plan.py:451-452)."""
    if isinstance(image, numpy.ndarray):
        if not image.flags["C_CONTIGUOUS"]:
            image = numpy.ascontiguousarray(image)              # plan.py:446-447
        return image.ctypes.data, 0, image.dtype, image.shape, image
    if hasattr(image, "data_ptr") and hasattr(image, "is_cuda"):   # torch.Tensor
        t = image if image.is_contiguous() else image.contiguous()
        np_dtype = numpy.dtype(str(t.dtype).replace("torch.", ""))
        if t.is_cuda:
            return t.data_ptr(), 1, np_dtype, tuple(t.shape), t
        arr = t.numpy()
        return arr.ctypes.data, 0, arr.dtype, arr.shape, arr
    cai = getattr(image, "__cuda_array_interface__", None)
    if cai is not None:
        if cai.get("strides") is not None:
            raise RuntimeError("device arrays must be C-contiguous")
        return cai["data"][0], 1, numpy.dtype(cai["typestr"]), tuple(cai["shape"]), image
    arr = numpy.ascontiguousarray(image)
    return arr.ctypes.data, 0, arr.dtype, arr.shape, arr


class _DeviceRecords(object):
    """Borrowed view of n 144-byte keypoint records in HBM (see SiftPlan.device_records)."""

    def __init__(self, ptr, n, owner):
        self._owner = owner
        self.count = int(n)
        self.shape = (n * 144,)
        self.__cuda_array_interface__ = {"shape": (n * 144,), "typestr": "|u1", "data": (ptr, False), "version": 2, "strides": None}


class _StageProfile(object):
    __slots__ = ("start", "end")

    def __init__(self, ns):
        self.start, self.end = 0, int(ns)


class StageEvent(object):
    """What ``SiftPlan.events`` holds beside a label under ``profile=True``: the reference appends ``(label, pyopencl event)``
    pairs (plan.py:331, 455, 522, 594) whose callers read ``evt.profile.end - evt.profile.start`` in nanoseconds
    (plan.py:838-846); this carries the hipEvent time of the stage the same way (``start`` is 0), and ``ms``."""
    __slots__ = ("profile", "ms")

    def __init__(self, ms):
        self.ms = float(ms)
        self.profile = _StageProfile(round(1e6 * self.ms))


class SiftPlan(object):
    """Plan to compute SIFT keypoints of images of one shape and type.

    ::

        siftp = sift_pyocl_amd.SiftPlan(img.shape, img.dtype, devicetype="GPU")
        kp = siftp.keypoints(img)

    ``kp`` is a numpy recarray with fields x, y, scale, angle and desc (128 x uint8), as in the
    reference (plan.py:110-115).

    Differences that are deliberate: the computation always runs on the MI355X selected by
    ``device`` (an integer HIP ordinal, or the reference's ``(platform, device)`` tuple whose
    second element is used); ``devicetype`` is accepted for compatibility -- the numerics are
    always those of the reference's CPU kernel variants (orientation_cpu / keypoints_cpu), which
    is what ``USE_CPU`` reports.  ``octave_max`` (extension) limits the number of octaves.
    """

    converter = {numpy.dtype(numpy.uint8): "u8_to_float",
                 numpy.dtype(numpy.uint16): "u16_to_float",
                 numpy.dtype(numpy.uint32): "u32_to_float",
                 numpy.dtype(numpy.uint64): "u64_to_float",
                 numpy.dtype(numpy.int32): "s32_to_float",
                 numpy.dtype(numpy.int64): "s64_to_float",
                 }
    sigmaRatio = 2.0 ** (1.0 / par.Scales)
    PIX_PER_KP = 10
    dtype_kp = numpy.dtype([('x', numpy.float32),
                            ('y', numpy.float32),
                            ('scale', numpy.float32),
                            ('angle', numpy.float32),
                            ('desc', (numpy.uint8, 128))
                            ])

    def __init__(self, shape=None, dtype=None, devicetype="CPU", template=None,
                 profile=False, device=None, PIX_PER_KP=None,
                 max_workgroup_size=None, context=None, init_sigma=None, octave_max=None):
        if init_sigma is None:
            init_sigma = par.InitSigma
        self._init_sigma = float(init_sigma)
        if template is not None:
            self.shape = tuple(template.shape)
            self.dtype = numpy.dtype(str(template.dtype).replace("torch.", ""))
        else:
            self.shape = tuple(shape)
            self.dtype = numpy.dtype(dtype)
        if len(self.shape) == 3:
            self.RGB = True
            self.shape = self.shape[:2]
        elif len(self.shape) == 2:
            self.RGB = False
        else:
            raise RuntimeError("Unable to process image of shape %s" % (tuple(self.shape,)))
        if PIX_PER_KP:
            self.PIX_PER_KP = int(PIX_PER_KP)
        if par.Scales != 3:
            raise RuntimeError("par.Scales is hard-wired to 3 in the kernels (as in image.cl:355)")
        self.profile = bool(profile)
        # profile=True: hipEvent bracket around every stage (log_profile(), as the reference);
        # profile="light": ONE event pair around the octave-0 blur launches and nothing else (what bench.py needs);
        # the first / last kernels are not bracketed at that level: kernel_times()["total_ms"] is 0
        self._profile_level = 0 if not profile else (1 if profile == "light" else 2)
        self.events = []
        self._sem = threading.Semaphore()
        self.scales = []     # octave sizes in XY order, as the reference
        self.procsize = []
        self.wgsize = []
        self.max_workgroup_size = max_workgroup_size or 4096   # accepted, unused
        self.ctx = context                                      # accepted, unused
        self._calc_scales()
        self._octave_limit = int(octave_max) if octave_max else 0
        if self._octave_limit:
            self.octave_max = min(self.octave_max, self._octave_limit)
        self._calc_memory()
        self.LOW_END = 0
        if isinstance(device, (tuple, list)):
            device = device[-1]
        if device is None:
            device = int(os.environ.get("SIFT_MI355X_DEVICE", os.environ.get("LOCAL_RANK", 0)))
        self.device = int(device)
        self.devicetype = "GPU"
        self.USE_CPU = (str(devicetype).upper() == "CPU")   # selects the CPU-variant numerics: always used here
        if self.RGB:
            if self.dtype != numpy.uint8:
                raise RuntimeError("invalid input format error (RGB needs uint8)")
            self._code = _lib.DTYPE_CODES["rgb8"]
        elif self.dtype == numpy.float64:
            self._code = _lib.DTYPE_CODES["float32"]        # host cast, as plan.py:457-463
        elif self.dtype == numpy.float32 or self.dtype in self.converter:
            self._code = _lib.DTYPE_CODES[self.dtype.name]
        else:
            raise RuntimeError("invalid input format error (%s)" % (str(self.dtype)))
        self._handle = C.c_void_p()
        L = _lib.lib()
        if L.siftmi_device_count() < 1:
            raise RuntimeError("sift_pyocl_amd needs a HIP device (MI355X); none is visible and there is no CPU fallback")
        #: par.DoubleImSize as the constructor saw it: the reference prepares the initial blur's taps from it here
        #: (plan.py:297-300) and looks them up again, by sigma, on every call (plan.py:534-539, 585)
        self._double_im = bool(par.DoubleImSize)
        self._params = self._current_params()
        self._par_key = (par.PeakThresh, par.EdgeThresh1, par.EdgeThresh, par.OriSigma, par.BorderDist, par.DoubleImSize)
        self._create(L)
        self.overflow = False
        #: results up to 8 MB come back as views of pinned host blocks written by the kernels themselves (no copy after
        #: the last kernel); a block returns to the library's pool when the array is dropped.  Set False to get plain
        #: numpy.empty arrays filled by a device-to-host copy (e.g. when thousands of results are kept alive).
        self.pinned_results = True
        self._last_n = 0
        self.debug = []

    def _create(self, L):
        """Allocate the device side (every buffer of the plan, as plan.py:268-306)."""
        _lib.check(L.siftmi_plan_create(self.shape[0], self.shape[1], self._code, self.device,
                                        C.byref(self._params), self._profile_level, C.byref(self._handle)))
        nbytes = C.c_int64()
        _lib.check(L.siftmi_plan_info(self._handle, None, None, C.byref(nbytes)))
        self.memory = int(nbytes.value)

    def _destroy(self, L, h):
        L.siftmi_plan_destroy(h)

    # ------------------------------------------------------------------ sizing (plan.py:213-266)
    def _calc_scales(self):
        shape = self.shape[-1::-1]
        self.scales = [tuple(numpy.int32(i) for i in shape)]
        min_size = 2 * par.BorderDist + 2
        while min(shape) > min_size:
            shape = tuple(numpy.int32(i // 2) for i in shape)
            self.scales.append(shape)
        self.scales.pop()
        self.octave_max = len(self.scales)
        for s in self.scales:
            wg = (min(nextpower(int(s[0])), 4096), 1)
            self.wgsize.append(wg)
            self.procsize.append(calc_size(s, wg))

    def _calc_memory(self):
        self.kpsize = max(1, int(self.shape[0] * self.shape[1] // self.PIX_PER_KP))
        self.red_size = nextpower(min(4096, math.sqrt(self.shape[0] * self.shape[1])))

    def gaussian_sizes(self):
        """[(sigma, taps)] of the initial blur (if any) and of the five per-octave blurs."""
        out = []
        cur = 1.0 if self._double_im else 0.5
        if self._init_sigma > cur:
            s = math.sqrt(self._init_sigma ** 2 - cur ** 2)
            out.append((s, kernel_size(s, True)))
        prev = self._init_sigma
        for _ in range(par.Scales + 2):
            inc = prev * math.sqrt(self.sigmaRatio ** 2 - 1.0)
            out.append((inc, kernel_size(inc, True)))
            prev *= self.sigmaRatio
        return out

    def _current_params(self):
        # par.DoubleImSize: all the reference does with it is to count the input as blurred by sigma 1.0 instead of 0.5
        # (plan.py:254, 297, 534) -- the image is never resampled.  A value that changed since the constructor asks for
        # an initial blur whose taps the reference has not prepared: its look-up by sigma fails (plan.py:585).
        dbl = bool(par.DoubleImSize)
        cur = 1.0 if dbl else 0.5
        if dbl != self._double_im and self._init_sigma > cur:
            raise KeyError("gaussian_%s" % math.sqrt(self._init_sigma ** 2 - cur ** 2))
        return _lib.Params(init_sigma=self._init_sigma,
                           peak_thresh=numpy.float32(par.PeakThresh),
                           edge_thresh0=numpy.float32(par.EdgeThresh1),
                           edge_thresh=numpy.float32(par.EdgeThresh),
                           ori_sigma=numpy.float32(par.OriSigma),
                           border_dist=int(par.BorderDist),
                           octave_max=self._octave_limit,
                           pix_per_kp=int(self.PIX_PER_KP), double_im_size=int(dbl))

    def __del__(self):
        h = getattr(self, "_handle", None)
        if h:
            try:
                self._destroy(_lib.lib(), h)
            except Exception:
                pass
            self._handle = None

    # ------------------------------------------------------------------ the hot path
    def keypoints(self, image):
        """Calculates the keypoints of the image

        :param image: 2D array (3D if RGB): numpy, or a device-resident torch tensor
        :return: numpy recarray of keypoints (x, y, scale, angle, desc[128])
        """
        self.reset_timer()
        with self._sem:
            t0 = time.time()
            ptr, is_dev, dtype, shape, keep = _pointer_of(image)
            assert tuple(shape[:2]) == tuple(self.shape)
            assert dtype in [self.dtype, numpy.float32]
            if dtype == numpy.float32 and len(shape) == 2:
                code = _lib.DTYPE_CODES["float32"]
            elif self.dtype == numpy.float64 and dtype == numpy.float64:
                if is_dev:
                    keep = keep.float()
                    ptr = keep.data_ptr()
                else:
                    keep = keep.astype(numpy.float32)
                    ptr = keep.ctypes.data
                code = _lib.DTYPE_CODES["float32"]
            elif len(shape) == 3 and dtype == numpy.uint8 and self.RGB:
                code = _lib.DTYPE_CODES["rgb8"]
            elif self.dtype in self.converter and len(shape) == 2:
                code = _lib.DTYPE_CODES[self.dtype.name]
            else:
                raise RuntimeError("invalid input format error (%s)" % (str(self.dtype)))
            L = _lib.lib()
            key = (par.PeakThresh, par.EdgeThresh1, par.EdgeThresh, par.OriSigma, par.BorderDist, par.DoubleImSize)
            if key != self._par_key:                 # `par` is read at call time, as in the reference
                params = self._current_params()
                _lib.check(L.siftmi_plan_set_params(self._handle, C.byref(params)))
                self._params, self._par_key = params, key
            n = C.c_int64(0)
            ovf = C.c_int32(0)
            # One call: the records land in an array sized from the previous frame's count (x1.5; untouched pages of
            # numpy.empty cost nothing), so the host does not come back to Python between the count and the copy.  If the
            # frame has more keypoints than guessed, the records are still on the device: fetch them into an exact array.
            cap = int(1.5 * self._last_n) + 256
            output = None
            if cap * 144 <= (8 << 20) and self.pinned_results:
                # pinned result array from the library's pool: the descriptor kernels write the records straight into
                # it (SIFTMI_OUT_PINNED), nothing is copied after the last kernel; it is recycled when the caller drops it.
                # Callers that keep many results alive accumulate page-locked memory: when the pool cannot grow any more
                # the call falls back to an ordinary array (one copy after the last kernel).
                try:
                    output = _lib.pinned_empty(cap, self.dtype_kp)
                except MemoryError:
                    output = None
            if output is not None:
                rc = L.siftmi_plan_keypoints(self._handle, ptr, code, is_dev, output.ctypes.data, 2, cap, C.byref(n), C.byref(ovf))
                _lib.check(rc, allow=(_lib.ECAPACITY,))
                count = n.value
                exact = rc == _lib.ECAPACITY
            elif cap * 144 <= (8 << 20):
                output = numpy.empty(cap, dtype=self.dtype_kp)
                rc = L.siftmi_plan_keypoints(self._handle, ptr, code, is_dev, output.ctypes.data, 0, cap, C.byref(n), C.byref(ovf))
                _lib.check(rc, allow=(_lib.ECAPACITY,))
                count = n.value
                exact = rc == _lib.ECAPACITY
            else:
                # large results: an over-sized array would be a fresh mmap (first-touch page faults on every call);
                # count first, then copy into an exactly sized array that the allocator can recycle
                _lib.check(L.siftmi_plan_keypoints(self._handle, ptr, code, is_dev, None, 0, 0, C.byref(n), C.byref(ovf)))
                exact = True
            if exact:
                total = C.c_int64(0)
                _lib.check(L.siftmi_plan_records_device(self._handle, C.byref(C.c_void_p()), C.byref(total)))
                count = total.value
                # Large lists too come back in a pinned block of the library's pool: a fresh numpy.empty of tens of MB is a
                # new mmap on every call -- page faults under the DMA, the runtime registers the pages with the GPU for the
                # copy, and when the caller drops the array the munmap invalidates that mapping, which stalls every queue of
                # the process for 20-70 ms (measured in LinearAlign.align, round 4).  Pool blocks are recycled instead.
                output = None
                if self.pinned_results and count:
                    try:
                        output = _lib.pinned_empty(count, self.dtype_kp)
                    except MemoryError:
                        output = None
                if output is None:
                    output = numpy.empty(count, dtype=self.dtype_kp)
                if count:
                    _lib.check(L.siftmi_plan_fetch(self._handle, output.ctypes.data, 0, 0, count))
            self._last_n = count
            if count * 144 <= (16 << 10) and output.nbytes >= (64 << 10) and not exact:
                output = output[:count].copy()      # a small result does not keep a 64 KiB page-locked block alive
            else:
                output = output[:count]
            self.overflow = bool(ovf.value)
            if self.overflow:
                logger.warning("Keypoint counter overflow: an octave needs more than %s entries, result cut to that per octave", self.kpsize)
            output = output.view(numpy.recarray)
            if self._profile_level == 2:         # (label, event) per stage of this call, as the reference's plan.events
                self.events = [(label, StageEvent(ms)) for label, ms in self._profile_lines()]
            del keep
            if logger.isEnabledFor(logging.INFO):
                logger.info("Execution time: %.3fms" % (1000 * (time.time() - t0)))
        return output

    __call__ = keypoints

    def set_option(self, name, value):
        """Tuning / diagnostic option of the device plan by name (``siftmi_plan_set_option`` in include/siftmi.h);
        results never depend on an option."""
        _lib.check(_lib.lib().siftmi_plan_set_option(self._handle, str(name).encode(), int(value)))

    def capacity(self):
        """(records the device list holds now, times a list has grown).  The lists start at ``kpsize`` entries -- the
        reference's capacity per octave (plan.py:243, 797-804) -- and grow when an image within that rule needs more."""
        rec, grows = C.c_int64(), C.c_int64()
        _lib.check(_lib.lib().siftmi_plan_capacity(self._handle, C.byref(rec), C.byref(grows)))
        return int(rec.value), int(grows.value)

    def tail_timeouts(self):
        """(images that ran again because the one-launch form of the small octaves timed out, whether the plan still uses that
        form).  The wait between its workgroups is bounded; a time-out costs one re-run and never a wrong result."""
        n, on = C.c_int64(), C.c_int32()
        _lib.check(_lib.lib().siftmi_plan_tail_timeouts(self._handle, C.byref(n), C.byref(on)))
        return int(n.value), bool(on.value)

    def device_records(self):
        """The records of the last keypoints() call where they lie on the device (no copy): an object with
        ``__cuda_array_interface__`` (uint8, n * 144 bytes), accepted by ``MatchPlan.match``.  Valid until the next
        call on this plan -- the reference's equivalent is matching ``pyopencl.array`` keypoints in place."""
        ptr = C.c_void_p()
        n = C.c_int64()
        _lib.check(_lib.lib().siftmi_plan_records_device(self._handle, C.byref(ptr), C.byref(n)))
        return _DeviceRecords(ptr.value or 0, int(n.value), self)

    def minmax(self):
        """(min, max) of the last processed image (buffers["min"], buffers["max"] in the reference)."""
        mn, mx = C.c_float(), C.c_float()
        _lib.check(_lib.lib().siftmi_plan_get_minmax(self._handle, C.byref(mn), C.byref(mx)))
        return mn.value, mx.value

    # ------------------------------------------------------------------ profiling (plan.py:826-857)
    def _profile_lines(self):
        buf = C.create_string_buffer(1 << 16)
        _lib.check(_lib.lib().siftmi_plan_profile(self._handle, buf, len(buf)))
        out = []
        for line in buf.value.decode().splitlines():
            label, ms = line.rsplit("\t", 1)
            out.append((label, float(ms)))
        return out

    def kernel_times(self):
        """dict(total_ms, blur_ms, blur_launches, blur_pixels) of the last call (profile=True; with profile="light" only
        the blur figures are measured and total_ms is 0)."""
        tot, blur = C.c_float(), C.c_float()
        nl, px = C.c_int32(), C.c_double()
        _lib.check(_lib.lib().siftmi_plan_last_kernel_ms(self._handle, C.byref(tot), C.byref(blur), C.byref(nl), C.byref(px)))
        out = dict(total_ms=tot.value, blur_ms=blur.value, blur_launches=nl.value, blur_pixels=px.value)
        _lib.check(_lib.lib().siftmi_plan_blur_ms(self._handle, 0, C.byref(blur), C.byref(nl), C.byref(px)))
        out.update(blur0_ms=blur.value, blur0_launches=nl.value, blur0_pixels=px.value)
        return out

    def profile_totals(self, reset=False):
        """Running totals of ``kernel_times()`` over every call since the last reset (profile="light"): dict(calls,
        total_ms, blur0_ms, blur0_launches, blur0_pixels); total_ms stays 0 under the light profile (no first / last events).  A benchmark loop reads this once after its timed region."""
        calls, nl = C.c_int64(), C.c_int64()
        tot, b0, px = C.c_double(), C.c_double(), C.c_double()
        _lib.check(_lib.lib().siftmi_plan_profile_totals(self._handle, int(bool(reset)), C.byref(calls), C.byref(tot), C.byref(b0),
                                                         C.byref(nl), C.byref(px)))
        return dict(calls=calls.value, total_ms=tot.value, blur0_ms=b0.value, blur0_launches=nl.value, blur0_pixels=px.value)

    def count_kp(self, output):
        """Print the number of keypoint per octave (plan.py:811-821): `output` = one (n, 4) array per octave,
        rows with column 1 == -1 are holes"""
        kpt = 0
        for octave, data in enumerate(output):
            if len(output) > 0:
                ksum = int((numpy.asarray(data)[:, 1] != -1.0).sum())
                kpt += ksum
                print("octave %i kp count %i/%i size %s ratio:%s" % (octave, ksum, self.kpsize, self.scales[octave],
                                                                    1000.0 * ksum / self.scales[octave][1] / self.scales[octave][0]))
        print("Found total %i guess %s pixels per keypoint" % (kpt, self.shape[0] * self.shape[1] / max(kpt, 1)))

    def debug_holes(self, label=""):
        """plan.py:823-824 prints the holes of the keypoint buffer; the lists of this build are appended through a device
        counter and have none."""
        print("%s %s" % (label, numpy.empty(0, dtype=numpy.int64)))

    def log_profile(self):
        """If profiling is on, print the device time of every stage of the last call."""
        t = orient = descr = 0.0
        if self.profile:
            for label, evt in (self.events or [(l, StageEvent(ms)) for l, ms in self._profile_lines()]):
                et = 1e-6 * (evt.profile.end - evt.profile.start)          # as plan.py:838-839
                print("%50s:\t%.3fms" % (label, et))
                t += et
                if "orient" in label:
                    orient += et
                if "descriptors" in label:
                    descr += et
        print("_" * 80)
        print("%50s:\t%.3fms" % ("Total execution time", t))
        print("%50s:\t%.3fms" % ("Total Orientation assignment", orient))
        print("%50s:\t%.3fms" % ("Total Descriptors", descr))

    def reset_timer(self):
        """plan.py:849-855: forget the events of the last call"""
        with self._sem:
            self.events = []


def demo():
    """plan.py:857-866 (scipy's sample image needs a download: a synthetic frame is used instead)"""
    rng = numpy.random.default_rng(0)
    from scipy.ndimage import gaussian_filter
    img = gaussian_filter(rng.random((512, 512)), 2.0).astype(numpy.float32)
    s = SiftPlan(template=img)
    print(s.keypoints(img))


if __name__ == "__main__":
    demo()
