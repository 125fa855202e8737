"""MatchPlan -- brute-force keypoint matching on one MI355X through libsiftmi.so.

Mirror of the reference's ``sift_pyocl.MatchPlan`` (sift-src/match.py:52-327): same constructor
keywords, ``match(nkp1, nkp2, raw_results=False)`` contract and record type.  The distance is the
reference's: L1 over the 128 descriptor bytes, best / second best with strict '<', pair kept iff
``dist2 != 0 and dist1 / dist2 < par.MatchRatio ** 2`` (matching_cpu.cl:57-109).
"""
import ctypes as C
import logging
import os
import threading

import numpy

from . import _lib
from .param import par
from .plan import _pointer_of

logger = logging.getLogger("sift.match")


class MatchPlan(object):
    """Plan to compare sets of SIFT keypoints and find common ones.

    ::

        mp = sift_pyocl_amd.MatchPlan()
        pairs = mp.match(kp1, kp2)                  # (m, 2) recarray of matching keypoints
        idx = mp.match(kp1, kp2, raw_results=True)  # (m, 2) int32 indices
    """
    dtype_kp = numpy.dtype([('x', numpy.float32),
                            ('y', numpy.float32),
                            ('scale', numpy.float32),
                            ('angle', numpy.float32),
                            ('desc', (numpy.uint8, 128))
                            ])

    def __init__(self, size=16384, devicetype="CPU", profile=False, device=None, max_workgroup_size=None,
                 roi=None, context=None):
        self.profile = bool(profile)
        self.events = []
        self.kpsize = int(size)
        self.max_workgroup_size = max_workgroup_size
        self.ctx = context
        if isinstance(device, (tuple, list)):
            device = device[-1]
        if device is None:
            device = int(os.environ.get("SIFT_MI355X_DEVICE", os.environ.get("LOCAL_RANK", 0)))
        self.device = int(device)
        self.devicetype = "GPU"
        self.USE_CPU = (str(devicetype).upper() == "CPU")
        self._sem = threading.Semaphore()
        self._handle = C.c_void_p()
        L = _lib.lib()
        if L.siftmi_device_count() < 1:
            raise RuntimeError("sift_pyocl_amd needs a HIP device (MI355X); none is visible and there is no CPU fallback")
        _lib.check(L.siftmi_match_create(self.kpsize, self.device, int(self.profile), C.byref(self._handle)))
        self.roi = None
        if roi is not None:
            self.set_roi(roi)

    def __del__(self):
        h = getattr(self, "_handle", None)
        if h:
            try:
                _lib.lib().siftmi_match_destroy(h)
            except Exception:
                pass
            self._handle = None

    ROI_MODES = {0: 0, None: 0, False: 0, "off": 0, 1: 1, True: 1, "reference": 1, "matching_valid": 1, 2: 2, "strict": 2}

    def match(self, nkp1, nkp2, raw_results=False, roi_mode=0, mutual=False):
        """Calculate the matching of 2 keypoint lists

        :param nkp1, nkp2: numpy 1D recarray of keypoints (or device tensors of 144-byte records)
        :param raw_results: if true return the 2D array of indexes of matching keypoints (not the actual keypoints)
        :param roi_mode: 0 (default) ignores the region of interest, exactly like the reference, whose ``match`` never
                         reaches its ``matching_valid`` kernel; "reference" / 1 runs that kernel's semantics literally
                         (matching_cpu.cl:136-199); "strict" / 2 drops every keypoint that is not on a non-zero pixel
        :param mutual: keep only pairs that are nearest neighbours in both directions (extension)
        """
        assert len(nkp1.shape) == 1
        assert len(nkp2.shape) == 1
        p1, dev1, n1, keep1 = self._records(nkp1)
        p2, dev2, n2, keep2 = self._records(nkp2)
        with self._sem:
            L = _lib.lib()
            if min(n1, n2) > self.kpsize:      # match.py:241-243
                self.kpsize = min(n1, n2)
            cap = max(1, self.kpsize)
            pairs = numpy.empty((cap, 2), dtype=numpy.int32)
            n = C.c_int64(0)
            total = C.c_int64(0)
            ratio = numpy.float32(par.MatchRatio * par.MatchRatio)
            mode = self.ROI_MODES[roi_mode]
            if mode and self.roi is None:
                raise RuntimeError("roi_mode=%r needs a region of interest (set_roi)" % (roi_mode,))
            _lib.check(L.siftmi_match_ex(self._handle, p1, n1, dev1, p2, n2, dev2, C.c_float(ratio), mode, int(bool(mutual)),
                                         pairs.ctypes.data, cap, C.byref(n), C.byref(total)), allow=(_lib.ECAPACITY,))
            size = int(n.value)
            if self.profile:             # match.py:226-263: (label, event) pairs of this call, appended until reset_timer()
                self.events += self._stage_events()
            match = pairs[:size].copy()
            if raw_results:
                result = match
            else:
                if dev1 or dev2:
                    raise RuntimeError("raw_results=False needs host keypoint arrays")
                result = numpy.recarray(shape=(size, 2), dtype=self.dtype_kp)
                result[:, 0] = keep1[match[:size, 0]]
                result[:, 1] = keep2[match[:size, 1]]
        return result

    __call__ = match

    def _records(self, kp):
        if isinstance(kp, numpy.ndarray):
            arr = numpy.ascontiguousarray(kp)
            if arr.dtype.itemsize != 144:
                raise RuntimeError("keypoints must be 144-byte (x, y, scale, angle, desc[128]) records")
            return arr.ctypes.data, 0, int(arr.shape[0]), arr
        ptr, is_dev, dtype, shape, keep = _pointer_of(kp)
        nbytes = int(numpy.prod(shape)) * dtype.itemsize
        if nbytes % 144:
            raise RuntimeError("device keypoint buffer is not a whole number of 144-byte records")
        return ptr, is_dev, nbytes // 144, keep

    STAGE_LABELS = ("copy H->D KP_1", "copy H->D KP_2", "matching", "copy D->H match")

    def _stage_events(self):
        """The reference's profiling events of one ``match`` call (match.py:226, 237, 261, 263) with the device time of each
        stage in place of the pyopencl event: ``evt.profile.end - evt.profile.start`` is nanoseconds, as there.  A stage that
        did not run (a device-resident list, no pair to copy back) has no entry -- the reference appends none either."""
        from .plan import StageEvent
        ms = (C.c_float * 4)()
        _lib.check(_lib.lib().siftmi_match_last_stage_ms(self._handle, ms))
        return [(label, StageEvent(v)) for label, v in zip(self.STAGE_LABELS, ms) if v >= 0.0]

    def log_profile(self):
        """Print the recorded stage times (the reference's classes share this loop: alignment.py:363-375, plan.py:832-846)"""
        t = 0.0
        if self.profile:
            for label, evt in self.events:
                et = 1e-6 * (evt.profile.end - evt.profile.start)
                print("%50s:\t%.3fms" % (label, et))
                t += et
        print("_" * 80)
        print("%50s:\t%.3fms" % ("Total execution time", t))

    def kernel_ms(self):
        ms = C.c_float()
        _lib.check(_lib.lib().siftmi_match_last_kernel_ms(self._handle, C.byref(ms)))
        return ms.value

    def reset_timer(self):
        with self._sem:
            self.events = []

    def set_roi(self, roi):
        """Defines the region of interest (match.py:312-320): 2D array, non zero where pixels are valid.  As in the
        reference it has no effect on ``match()`` unless ``roi_mode`` is given."""
        with self._sem:
            self.roi = numpy.ascontiguousarray(roi, numpy.int8)
            if self.roi.ndim != 2:
                raise RuntimeError("the region of interest must be a 2D array")
            _lib.check(_lib.lib().siftmi_match_set_roi(self._handle, self.roi.ctypes.data, self.roi.shape[1], self.roi.shape[0]))

    def unset_roi(self):
        """Unset the region of interest (match.py:322-327)"""
        with self._sem:
            self.roi = None
            _lib.check(_lib.lib().siftmi_match_set_roi(self._handle, None, 0, 0))
