"""sift_pyocl_amd -- MI355X-native SIFT keypoints and matching.

Drop-in for the hot path of pierrepaleo/sift_pyocl: ``SiftPlan.keypoints()``,
``MatchPlan.match()`` and ``LinearAlign.align()`` (sift-src/__init__.py:29-33 exports the same names) over hand-written HIP
kernels for gfx950 reached through a C ABI (include/siftmi.h).
"""
version = "0.1"
from .param import par
from .plan import SiftPlan
from .match import MatchPlan
from .alignment import LinearAlign
from .batch import BatchPlan

__all__ = ["par", "SiftPlan", "MatchPlan", "LinearAlign", "BatchPlan", "version"]
