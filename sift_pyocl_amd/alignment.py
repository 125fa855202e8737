"""LinearAlign -- register images onto a reference frame with an affine map, on one MI355X.

Drop-in for the reference's ``sift_pyocl.LinearAlign`` (sift-src/alignment.py:76-360): constructor keywords,
the ``align(img, shift_only, return_all, double_check, relative, orsa)`` contract and the keys of the
``return_all`` dictionary are the reference's; the body is this package's own.  Device work per aligned frame:
``SiftPlan.keypoints`` -> ``MatchPlan.match`` (both keypoint lists stay in HBM) -> the affine warp kernel
(the arithmetic of openCL/transform.cl) through ``siftmi_plan_transform`` (include/siftmi.h).  The frame
uploaded for ``keypoints()`` stays staged on the device and is the one the warp reads.

Behavioural notes (deliberate):
  * the affine fit is a float64 least-squares solve (``utils.affine_least_squares``); the reference snapshot's
    ``utils.matching_correction`` stops before solving (utils.py:156-189);
  * the warp writes the whole ``outshape`` (the reference launches one work-item per *input* pixel and leaves
    the ``extra`` margin of its output undefined);
  * ``orsa=True`` needs the third-party ``feature`` module, exactly as in the reference; absent -> warning.
"""
import ctypes as C
import logging
import os
import threading

import numpy

from . import _lib
from .match import MatchPlan
from .plan import SiftPlan, StageEvent
from .utils import affine_least_squares, matching_correction  # noqa: F401  (matching_correction: reference API)

logger = logging.getLogger("sift.alignment")
try:
    import feature
except ImportError:
    feature = None

#: a fit needs three correspondences per degree of freedom of the affine map (alignment.py:262)
MIN_MATCHES_AFFINE = 3 * 6
#: rejection threshold of ``double_check`` in standard deviations (alignment.py:291-293)
OUTLIER_SIGMAS = 4


def arrow_start(kplist):
    """Tip of the scale/orientation arrow of every keypoint (reference helper, alignment.py:60-67):
    (x + scale*cos(angle), y + scale*sin(angle))."""
    length, theta = kplist.scale, kplist.angle
    return kplist.x + length * numpy.cos(theta), kplist.y + length * numpy.sin(theta)


def transform_pts(matrix, offset, x, y):
    """Apply the (y, x)-ordered affine map of the warp kernel to points (reference helper, alignment.py:70-73).
    Sums are formed in the reference's order so the float results are the same."""
    m = numpy.asarray(matrix)
    new_x = -offset[1] + y * m[1, 0] + x * m[1, 1]
    new_y = -offset[0] + x * m[0, 1] + y * m[0, 0]
    return new_x, new_y


def _as_yx_pair(extra):
    """``extra`` margin as a (y, x) tuple of ints; a scalar applies to both axes"""
    if hasattr(extra, "__len__"):
        return tuple(int(v) for v in extra[:2])
    return (int(extra),) * 2


def _zscore_flags(values, limit=OUTLIER_SIGMAS):
    """1 where a sample lies more than `limit` standard deviations from the mean (NaN z-scores compare False)"""
    with numpy.errstate(divide="ignore", invalid="ignore"):
        return (abs((values - values.mean()) / values.std()) > limit).astype(numpy.int8)


class LinearAlign(object):
    """Align images on a reference image based on an affine transformation (bi-linear + offset)"""

    def __init__(self, image, devicetype="GPU", profile=False, device=None, max_workgroup_size=None,
                 ROI=None, extra=0, context=None, init_sigma=None):
        """
        :param image: reference image on which the other images are aligned (2-D float-able, or H x W x 3 uint8)
        :param devicetype, max_workgroup_size, context: accepted for source compatibility with the reference, unused
        :param profile: collect kernel timings (``log_profile``)
        :param device: HIP device ordinal (default: SIFT_MI355X_DEVICE, LOCAL_RANK or 0)
        :param ROI: boolean mask; reference keypoints outside it are dropped
        :param extra: margin added around the output, an integer or a (y, x) pair
        :param init_sigma: blur width of the initial smoothing (default ``par.InitSigma`` = 1.6)
        """
        ndim = len(image.shape)
        if ndim not in (2, 3):
            raise RuntimeError("Unable to process image of shape %s" % (tuple(image.shape,)))
        self.RGB = ndim == 3
        self.shape = tuple(int(v) for v in image.shape[:2])
        self.extra = _as_yx_pair(extra)
        self.outshape = (self.shape[0] + 2 * self.extra[0], self.shape[1] + 2 * self.extra[1])
        self.profile = bool(profile)
        self.events = []
        self.ref = numpy.ascontiguousarray(image, numpy.float32)
        self.ROI = ROI
        self.ctx = context
        self.devicetype = "GPU"
        self.max_workgroup_size = max_workgroup_size
        if isinstance(device, (tuple, list)):       # the reference's (platform, device) pair: keep the device index
            device = device[-1]
        if device is None:
            device = os.environ.get("SIFT_MI355X_DEVICE", os.environ.get("LOCAL_RANK", 0))
        self.device = int(device)
        self.sift = SiftPlan(template=image, device=self.device, profile=self.profile, init_sigma=init_sigma)
        self.match = MatchPlan(device=self.device, profile=self.profile)
        self.fill_value = 0
        self.sem = threading.Semaphore()
        self.relative_transfo = None
        self.last_transform_ms = 0.0
        self._ref_dev = None
        self._set_reference(self.sift.keypoints(image))

    # ------------------------------------------------------------------ reference keypoints
    def _set_reference(self, kp):
        """Install `kp` (ROI-filtered) as the reference list and keep a copy resident on the device
        (the reference's buffers["ref_kp_gpu"], alignment.py:155-157)."""
        self.ref_kp = self._mask(kp)
        self._ref_heads = None               # dense (n, 4) copy of (x, y, scale, angle), built on first use
        self._upload_ref()

    def _upload_ref(self):
        """Device copy of the reference keypoints.  torch is only the allocator here: without it, or without a
        visible device, the host list is handed to every match() call instead."""
        self._ref_dev = None
        if not self.ref_kp.size:
            return
        try:
            import torch
            if not torch.cuda.is_available():
                return
            raw = numpy.ascontiguousarray(self.ref_kp).view(numpy.uint8).reshape(-1)
            self._ref_dev = torch.from_numpy(raw.copy()).to("cuda:%d" % self.device)
        except Exception as err:  # noqa: BLE001 -- any allocator problem: fall back to the host list
            logger.debug("reference keypoints stay on the host (%s)", err)
            self._ref_dev = None

    def _mask(self, kp):
        if self.ROI is None:
            return kp
        col = numpy.round(kp.x).astype(numpy.int32)
        row = numpy.round(kp.y).astype(numpy.int32)
        inside = self.ROI[(row, col)].astype(bool)
        logger.warning("Reducing keypoint list from %i to %i because of the ROI" % (kp.size, inside.sum()))
        return kp[inside]

    # ------------------------------------------------------------------ device warp
    def transform(self, matrix, offset, image=None, fill=None, mode=1):
        """Affine warp on the device (transform.cl).  image=None warps the image staged by the last keypoints()."""
        matrix = numpy.ascontiguousarray(matrix, numpy.float32).reshape(4)
        offset = numpy.ascontiguousarray(offset, numpy.float32).reshape(2)
        if fill is None:
            fill = self.sift.minmax()[0]
        # the result image lives in a pinned block of the library's pool (recycled when the caller drops it): the copy out
        # of HBM runs at the link's rate, and no 64 MB mmap / munmap pair per aligned frame (see SiftPlan.keypoints)
        oshape, odtype = (self.outshape + (3,), numpy.uint8) if self.RGB else (self.outshape, numpy.float32)
        # (sift.pinned_results = False opts out, as for keypoints(); beyond the pool's limit -- _lib.pinned_pool -- the call gets an
        # ordinary array as well: a stack-alignment loop that keeps every aligned frame must not pin the whole stack)
        out = None
        if getattr(self.sift, "pinned_results", True):
            try:
                out = _lib.pinned_empty(int(numpy.prod(oshape)), odtype).reshape(oshape)
            except MemoryError:
                out = None
        if out is None:
            out = numpy.empty(oshape, odtype)
        ptr = None
        if image is not None:
            image = numpy.ascontiguousarray(image, numpy.uint8 if self.RGB else numpy.float32)
            assert image.shape[:2] == self.shape
            ptr = image.ctypes.data
        ms = C.c_double(0)
        _lib.check(_lib.lib().siftmi_plan_transform(self.sift._handle, ptr, 0, 3 if self.RGB else 1, out.ctypes.data, 0,
                                                    self.outshape[1], self.outshape[0], matrix.ctypes.data, offset.ctypes.data,
                                                    C.c_float(fill), int(mode), C.byref(ms)))
        self.last_transform_ms = ms.value
        if self.profile:
            self.events.append(("transform", StageEvent(ms.value)))
        return out

    # ------------------------------------------------------------------ host-side estimation
    @staticmethod
    def _xysa(kp):
        """(n, 4) float32 view of (x, y, scale, angle) of a keypoint array (no copy for contiguous records)"""
        a = numpy.ascontiguousarray(kp)
        return a.view(numpy.uint8).reshape(a.shape[0], 144)[:, :16].view(numpy.float32)

    @staticmethod
    def _affine(g0, g1):
        """Least-squares affine map g0 -> g1, returned in the kernel's (y, x) ordering: the six coefficients
        (a..f) of matching_correction are unpacked as in alignment.py:279-283."""
        a, b, c, d, e, f = affine_least_squares(g0[:, 0], g0[:, 1], g1[:, 0], g1[:, 1])[:6]
        return numpy.array([[e, d], [b, a]], dtype=numpy.float32), numpy.array([f, c], dtype=numpy.float32)

    @staticmethod
    def _median_shift(g0, g1):
        """Translation-only estimate: identity matrix and the median displacement, (dy, dx) order (alignment.py:268-271)"""
        delta = g1[:, :2] - g0[:, :2]
        return numpy.identity(2, dtype=numpy.float32), numpy.array([numpy.median(delta[:, 1]), numpy.median(delta[:, 0])], numpy.float32)

    @staticmethod
    def _inliers(g0, g1):
        """double_check (alignment.py:284-296): a pair is an outlier when its displacement length, its angle
        difference or its log scale ratio lies more than OUTLIER_SIGMAS standard deviations from the mean.
        Returns the boolean keep mask, or None when nothing is to be rejected."""
        shift = g1[:, :2] - g0[:, :2]
        votes = _zscore_flags(numpy.sqrt(shift[:, 0] * shift[:, 0] + shift[:, 1] * shift[:, 1]))
        votes = votes + _zscore_flags(g1[:, 3] - g0[:, 3])
        votes = votes + _zscore_flags(numpy.log(g1[:, 2] / g0[:, 2]))
        rejected = votes.sum()
        if rejected > 0 and not numpy.isinf(rejected):
            return votes == 0
        return None

    def _chain_relative(self, matrix, offset):
        """relative=True: compose this frame's map with the maps of the previous frames (homogeneous 3x3 product,
        newest on the left, alignment.py:309-318) and return the accumulated matrix / offset."""
        step = numpy.identity(3, dtype=numpy.float64)
        step[:2, :2] = matrix
        step[:2, 2] = offset
        self.relative_transfo = step if self.relative_transfo is None else numpy.dot(step, self.relative_transfo)
        return (numpy.ascontiguousarray(self.relative_transfo[:2, :2], dtype=numpy.float32),
                numpy.ascontiguousarray(self.relative_transfo[:2, 2], dtype=numpy.float32))

    # ------------------------------------------------------------------ public entry
    def align(self, img, shift_only=False, return_all=False, double_check=False, relative=False, orsa=False):
        """Align `img` on the reference image.

        :param img: image to align (same shape as the reference)
        :param shift_only: estimate a translation only (also the fallback below 18 matches)
        :param return_all: return a dict with result / keypoint / matching / offset / matrix / rms instead of the image
        :param double_check: re-fit after rejecting 4-sigma outliers in displacement, angle and scale
        :param relative: make this frame the reference of the next one and accumulate the transformations
        :param orsa: filter the matches with ``feature.sift_orsa`` when that module is importable
        :return: the aligned image, the dict, or None when no keypoint matches
        """
        logger.debug("ref_keypoints: %s" % self.ref_kp.size)
        data = numpy.ascontiguousarray(img, numpy.uint8 if self.RGB else numpy.float32)
        with self.sem:
            kp = self.sift.keypoints(data)          # uploads `data`; it stays staged on the device for the warp
            logger.debug("mod image keypoints: %s" % kp.size)
            # both lists are matched where they lie in HBM: the reference list uploaded once, the new one still in the plan
            ref_list = self.ref_kp if self._ref_dev is None else self._ref_dev
            pairs = self.match.match(ref_list, self.sift.device_records() if kp.size else kp, raw_results=True)
            n_pairs = pairs.shape[0]
            if n_pairs == 0:
                logger.warning("No matching keypoints")
                return None
            # Only (x, y, scale, angle) of the matched keypoints enter the fit: 16 bytes per keypoint are gathered, not
            # the 144-byte records; the reference's `matching` recarray (alignment.py:254-259) is only materialised for
            # ORSA and for return_all.
            ref_used = self.ref_kp
            # (the 16-byte heads are first packed into a dense (n, 4) array -- one strided pass -- and indexed there: a fancy
            # index straight into the 144-byte records took 13 ms for 2 x 195 k pairs, this takes 3)
            if self._ref_heads is None:
                self._ref_heads = numpy.ascontiguousarray(self._xysa(ref_used))
            g0 = self._ref_heads[pairs[:, 0]]
            g1 = numpy.ascontiguousarray(self._xysa(kp))[pairs[:, 1]]

            def matched_records():
                both = numpy.recarray(shape=pairs.shape, dtype=MatchPlan.dtype_kp)
                both[:, 0] = ref_used[pairs[:, 0]]
                both[:, 1] = kp[pairs[:, 1]]
                return both

            matching = None
            if orsa and feature is None:
                logger.warning("feature is not available. No ORSA filtering")
            elif orsa:
                matching = feature.sift_orsa(matched_records(), self.shape, 1)
                g0, g1 = self._xysa(numpy.ascontiguousarray(matching[:, 0])), self._xysa(numpy.ascontiguousarray(matching[:, 1]))

            enough = n_pairs >= MIN_MATCHES_AFFINE
            if shift_only or not enough:
                (logger.debug if shift_only else logger.warning)("Shift Only mode: Common keypoints: %s" % n_pairs)
                matrix, offset = self._median_shift(g0, g1)
            else:
                logger.debug("Common keypoints: %s" % n_pairs)
                matrix, offset = self._affine(g0, g1)
            if double_check and enough:
                logger.warning("Validating keypoints, %s,%s" % (matrix, offset))
                keep = self._inliers(g0, g1)
                if keep is not None:
                    matrix, offset = self._affine(g0[keep], g1[keep])
            if relative:
                self._set_reference(kp)             # this frame becomes the reference of the next one
                matrix, offset = self._chain_relative(matrix, offset)
            result = self.transform(matrix, offset, image=None, fill=self.sift.minmax()[0], mode=1)

        if not return_all:
            return result
        # residual of the fitted map on the matched keypoints, in pixels (alignment.py:349-351)
        # (written out instead of numpy.dot(matrix, src): a threaded BLAS has no business in a 2 x 2 product, see utils.py)
        sy, sx = g0[:, 1], g0[:, 0]
        ry = matrix[0, 0] * sy + matrix[0, 1] * sx + offset[0] - g1[:, 1]
        rx = matrix[1, 0] * sy + matrix[1, 1] * sx + offset[1] - g1[:, 0]
        rms = numpy.sqrt((ry * ry + rx * rx).mean())
        if matching is None:
            matching = matched_records()
        return {"result": result, "keypoint": kp, "matching": matching, "offset": offset, "matrix": matrix, "rms": rms}

    __call__ = align

    def log_profile(self):
        """Print the timing of every recorded device call (profile=True)"""
        if not self.profile:
            return
        total = 0.0
        # alignment.py:363-375 walks self.events; here the plans the aligner drives keep the events of their own stages
        # (the reference's aligner shares one queue with them and sees only its own three): all of them are listed
        for e in list(self.events) + list(self.sift.events) + list(self.match.events):
            if "__len__" in dir(e) and len(e) >= 2:
                et = 1e-6 * (e[1].profile.end - e[1].profile.start)
                print("%50s:\t%.3fms" % (e[0], et))
                total += et
        print("_" * 80)
        print("%50s:\t%.3fms" % ("Total execution time", total))
