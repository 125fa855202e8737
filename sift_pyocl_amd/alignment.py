"""LinearAlign -- align images on a reference image with an affine transformation, on one MI355X.

Mirror of the reference's ``sift_pyocl.LinearAlign`` (sift-src/alignment.py:76-360): same constructor
keywords, ``align(img, shift_only, return_all, double_check, relative, orsa)`` contract and result
dictionary.  The device work is SiftPlan.keypoints -> MatchPlan.match -> the affine warp kernel
(openCL/transform.cl) reached through ``siftmi_plan_transform`` (include/siftmi.h); the image uploaded for
keypoints() stays staged on the device and is the one the warp reads, as the reference's buffers["input"].

Differences, all deliberate:
  * ``utils.matching_correction`` is completed with a least-squares solve (the reference snapshot's function
    ends before solving, see utils.py) -- so the affine branch works;
  * the warp kernel covers the whole ``outshape`` (the reference launches one work-item per *input* pixel
    and leaves the ``extra`` margin of its output buffer uninitialised);
  * ``orsa`` needs the third-party ``feature`` module exactly as in the reference; absent -> warning.
"""
import ctypes as C
import logging
import os
import threading

import numpy

from . import _lib
from .match import MatchPlan
from .plan import SiftPlan
from .utils import affine_least_squares, matching_correction  # noqa: F401  (matching_correction: reference API)

logger = logging.getLogger("sift.alignment")
try:
    import feature
except ImportError:
    feature = None


def arrow_start(kplist):
    """alignment.py:60-67"""
    angle_ref = kplist.angle
    scale_ref = kplist.scale
    x_ref2 = kplist.x + scale_ref * numpy.cos(angle_ref)
    y_ref2 = kplist.y + scale_ref * numpy.sin(angle_ref)
    return x_ref2, y_ref2


def transform_pts(matrix, offset, x, y):
    """alignment.py:70-73"""
    nx = -offset[1] + y * matrix[1, 0] + x * matrix[1, 1]
    ny = -offset[0] + x * matrix[0, 1] + y * matrix[0, 0]
    return nx, ny


class LinearAlign(object):
    """Align images on a reference image based on an afine transformation (bi-linear + offset)"""

    def __init__(self, image, devicetype="GPU", profile=False, device=None, max_workgroup_size=None,
                 ROI=None, extra=0, context=None, init_sigma=None):
        """
        :param image: reference image on which other image should be aligned
        :param devicetype, max_workgroup_size, context: accepted for source compatibility, unused
        :param profile: collect kernel timings
        :param device: HIP device ordinal (default: LOCAL_RANK or 0)
        :param ROI: boolean mask of the region where reference keypoints are kept
        :param extra: extra space around the image, an integer or a 2-tuple in YX convention
        :param init_sigma: bluring width, you should have good reasons to modify the 1.6 default value...
        """
        self.profile = bool(profile)
        self.events = []
        self.ref = numpy.ascontiguousarray(image, numpy.float32)
        self.shape = image.shape
        if len(self.shape) == 3:
            self.RGB = True
            self.shape = self.shape[:2]
        elif len(self.shape) == 2:
            self.RGB = False
        else:
            raise RuntimeError("Unable to process image of shape %s" % (tuple(self.shape,)))
        self.shape = tuple(int(i) for i in self.shape)
        if "__len__" not in dir(extra):
            self.extra = (int(extra), int(extra))
        else:
            self.extra = tuple(int(i) for i in extra[:2])
        self.outshape = tuple(i + 2 * j for i, j in zip(self.shape, self.extra))
        self.ROI = ROI
        self.ctx = context
        if isinstance(device, (tuple, list)):
            device = device[-1]
        if device is None:
            device = int(os.environ.get("SIFT_MI355X_DEVICE", os.environ.get("LOCAL_RANK", 0)))
        self.device = int(device)
        self.devicetype = "GPU"
        self.max_workgroup_size = max_workgroup_size
        self.sift = SiftPlan(template=image, device=self.device, profile=self.profile, init_sigma=init_sigma)
        self.ref_kp = self._mask(self.sift.keypoints(image))
        self.match = MatchPlan(device=self.device, profile=self.profile)
        self._ref_dev = None
        self._upload_ref()
        self.fill_value = 0
        self.sem = threading.Semaphore()
        self.relative_transfo = None
        self.last_transform_ms = 0.0

    def _upload_ref(self):
        """Reference keypoints resident on the device (alignment.py:155-157: buffers["ref_kp_gpu"]); needs torch for
        the allocation, otherwise the host list is sent with every match."""
        self._ref_dev = None
        try:
            import torch
            if self.ref_kp.size:
                raw = numpy.ascontiguousarray(self.ref_kp).view(numpy.uint8).reshape(-1)
                self._ref_dev = torch.from_numpy(raw.copy()).to("cuda:%d" % self.device)
        except ImportError:
            pass

    def _mask(self, kp):
        if self.ROI is None:
            return kp
        kpx = numpy.round(kp.x).astype(numpy.int32)
        kpy = numpy.round(kp.y).astype(numpy.int32)
        masked = self.ROI[(kpy, kpx)].astype(bool)
        logger.warning("Reducing keypoint list from %i to %i because of the ROI" % (kp.size, masked.sum()))
        return kp[masked]

    def transform(self, matrix, offset, image=None, fill=None, mode=1):
        """Affine warp on the device (transform.cl).  image=None warps the image staged by the last keypoints()."""
        matrix = numpy.ascontiguousarray(matrix, numpy.float32).reshape(4)
        offset = numpy.ascontiguousarray(offset, numpy.float32).reshape(2)
        if fill is None:
            fill = self.sift.minmax()[0]
        if self.RGB:
            out = numpy.empty(self.outshape + (3,), numpy.uint8)
        else:
            out = numpy.empty(self.outshape, numpy.float32)
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
            self.events.append(("transform", ms.value))
        return out

    def align(self, img, shift_only=False, return_all=False, double_check=False, relative=False, orsa=False):
        """
        Align image on reference image

        :param img: numpy array containing the image to align to reference
        :param return_all: return in addition ot the image, keypoints, matching keypoints, and transformations as a dict
        :param relative: update reference keypoints with those from current image to perform relative alignment
        :return: aligned image or all informations
        """
        logger.debug("ref_keypoints: %s" % self.ref_kp.size)
        if self.RGB:
            data = numpy.ascontiguousarray(img, numpy.uint8)
        else:
            data = numpy.ascontiguousarray(img, numpy.float32)
        with self.sem:
            kp = self.sift.keypoints(data)          # uploads `data`; it stays staged on the device for the warp
            logger.debug("mod image keypoints: %s" % kp.size)
            # both lists are matched where they lie in HBM: the reference list uploaded once, the new one still in the plan
            raw_matching = self.match.match(self._ref_dev if self._ref_dev is not None else self.ref_kp,
                                            self.sift.device_records() if kp.size else kp, raw_results=True)
            len_match = raw_matching.shape[0]
            if len_match == 0:
                logger.warning("No matching keypoints")
                return
            # Only (x, y, scale, angle) of the matched keypoints enter the fit: gather those 16 bytes per keypoint
            # instead of the 144-byte records; the `matching` recarray of the reference (alignment.py:254-259) is
            # built lazily, for ORSA and for return_all.
            g0 = self._xysa(self.ref_kp)[raw_matching[:, 0]]
            g1 = self._xysa(kp)[raw_matching[:, 1]]
            matching = None
            ref_kp_used = self.ref_kp

            def full_matching():
                m = numpy.recarray(shape=raw_matching.shape, dtype=MatchPlan.dtype_kp)
                m[:, 0] = ref_kp_used[raw_matching[:, 0]]
                m[:, 1] = kp[raw_matching[:, 1]]
                return m

            if orsa:
                if feature:
                    matching = feature.sift_orsa(full_matching(), self.shape, 1)
                    g0 = numpy.stack([matching[:, 0].x, matching[:, 0].y, matching[:, 0].scale, matching[:, 0].angle], axis=1)
                    g1 = numpy.stack([matching[:, 1].x, matching[:, 1].y, matching[:, 1].scale, matching[:, 1].angle], axis=1)
                else:
                    logger.warning("feature is not available. No ORSA filtering")

            if (len_match < 3 * 6) or (shift_only):  # 3 points per DOF
                if shift_only:
                    logger.debug("Shift Only mode: Common keypoints: %s" % len_match)
                else:
                    logger.warning("Shift Only mode: Common keypoints: %s" % len_match)
                dx = g1[:, 0] - g0[:, 0]
                dy = g1[:, 1] - g0[:, 1]
                matrix = numpy.identity(2, dtype=numpy.float32)
                offset = numpy.array([+numpy.median(dy), +numpy.median(dx)], numpy.float32)
            else:
                logger.debug("Common keypoints: %s" % len_match)
                matrix, offset = self._affine(g0, g1)
            if double_check and (len_match >= 3 * 6):
                logger.warning("Validating keypoints, %s,%s" % (matrix, offset))
                dx = g1[:, 0] - g0[:, 0]
                dy = g1[:, 1] - g0[:, 1]
                dangle = g1[:, 3] - g0[:, 3]
                dscale = numpy.log(g1[:, 2] / g0[:, 2])
                distance = numpy.sqrt(dx * dx + dy * dy)
                outlayer = numpy.zeros(distance.shape, numpy.int8)
                outlayer += abs((distance - distance.mean()) / distance.std()) > 4
                outlayer += abs((dangle - dangle.mean()) / dangle.std()) > 4
                outlayer += abs((dscale - dscale.mean()) / dscale.std()) > 4
                outlayersum = outlayer.sum()
                if outlayersum > 0 and not numpy.isinf(outlayersum):
                    keep = outlayer == 0
                    matrix, offset = self._affine(g0[keep], g1[keep])
            if relative:  # update stable part to perform a relative alignment
                self.ref_kp = self._mask(kp)
                self._upload_ref()
                transfo = numpy.zeros((3, 3), dtype=numpy.float64)
                transfo[:2, :2] = matrix
                transfo[0, 2] = offset[0]
                transfo[1, 2] = offset[1]
                transfo[2, 2] = 1
                if self.relative_transfo is None:
                    self.relative_transfo = transfo
                else:
                    self.relative_transfo = numpy.dot(transfo, self.relative_transfo)
                matrix = numpy.ascontiguousarray(self.relative_transfo[:2, :2], dtype=numpy.float32)
                offset = numpy.ascontiguousarray(self.relative_transfo[:2, 2], dtype=numpy.float32)
            result = self.transform(matrix, offset, image=None, fill=self.sift.minmax()[0], mode=1)

        if return_all:
            corr = numpy.dot(matrix, numpy.vstack((g0[:, 1], g0[:, 0]))).T + offset.T - numpy.vstack((g1[:, 1], g1[:, 0])).T
            rms = numpy.sqrt((corr * corr).sum(axis=-1).mean())
            if matching is None:
                matching = full_matching()
            return {"result": result, "keypoint": kp, "matching": matching, "offset": offset, "matrix": matrix, "rms": rms}
        return result

    __call__ = align

    @staticmethod
    def _xysa(kp):
        """(n, 4) float32 view of (x, y, scale, angle) of a keypoint array (no copy for contiguous records)"""
        a = numpy.ascontiguousarray(kp)
        return a.view(numpy.uint8).reshape(a.shape[0], 144)[:, :16].view(numpy.float32)

    @staticmethod
    def _affine(g0, g1):
        """alignment.py:279-283: (a..f) of matching_correction -> the (y, x)-ordered matrix / offset of the kernel"""
        t = affine_least_squares(g0[:, 0], g0[:, 1], g1[:, 0], g1[:, 1])
        offset = numpy.array([t[5], t[2]], dtype=numpy.float32)
        matrix = numpy.empty((2, 2), dtype=numpy.float32)
        matrix[0, 0], matrix[0, 1] = t[4], t[3]
        matrix[1, 0], matrix[1, 1] = t[1], t[0]
        return matrix, offset

    def log_profile(self):
        """If we are in debugging mode, prints out all timing for every single kernel call"""
        t = 0.0
        if self.profile:
            for name, ms in self.events:
                print("%50s:\t%.3fms" % (name, ms))
                t += ms
            print("_" * 80)
            print("%50s:\t%.3fms" % ("Total execution time", t))
