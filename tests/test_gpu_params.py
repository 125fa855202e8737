"""GPU parity with non-default `par` values: the reference reads sift_pyocl.param.par at CALL time
(sift-src/plan.py:255-259 reads par.* inside keypoints() / _one_octave, plan.py:630-634), so mutating it between two
calls on the same plan must take effect -- through siftmi_plan_set_params -- and the result must equal the oracle run
with the same thresholds."""
import numpy as np
import pytest

from util import assert_same_keypoints, smooth_noise

pytestmark = pytest.mark.gpu

CASES = [
    dict(),                                                      # defaults
    dict(PeakThresh=255.0 * 0.02 / 3.0),                         # lower contrast threshold: more keypoints
    dict(PeakThresh=255.0 * 0.08 / 3.0, EdgeThresh=0.03, EdgeThresh1=0.04),
    dict(OriSigma=1.0),                                          # narrower orientation window
    dict(OriSigma=2.0, EdgeThresh1=0.12),
    dict(BorderDist=9),                                          # detection border only; octave list stays the plan's
    dict(BorderDist=7, PeakThresh=255.0 * 0.03 / 3.0, OriSigma=1.25),
]


def test_par_mutation_between_calls(siftlib, oracle):
    import sift_pyocl_amd as sp
    from sift_pyocl_amd.param import par
    saved = dict(par)
    img = smooth_noise((333, 402), seed=17, sigma=2.0)
    plan = sp.SiftPlan(template=img)
    counts = []
    try:
        for case in CASES + [dict()]:                            # ... and back to the defaults on the same plan
            par.update(saved)
            par.update(case)
            want = oracle.default_params()
            want.peak_thresh = np.float32(par.PeakThresh)
            want.edge_thresh0 = np.float32(par.EdgeThresh1)
            want.edge_thresh = np.float32(par.EdgeThresh)
            want.ori_sigma = np.float32(par.OriSigma)
            want.border_dist = int(par.BorderDist)
            got = plan.keypoints(img)
            assert_same_keypoints(got, oracle.keypoints(img, par=want), "par %r" % (case,))
            counts.append(len(got))
    finally:
        par.update(saved)
    assert counts[0] == counts[-1] and len(set(counts)) >= 5, counts   # the knobs really changed the result


def test_unknown_option_and_fixed_fields(siftlib):
    import sift_pyocl_amd as sp
    plan = sp.SiftPlan(shape=(128, 160), dtype=np.float32)
    with pytest.raises(RuntimeError):
        plan.set_option("no_such_option", 1)
    plan.set_option("overlap", 0)
    plan.set_option("overlap", 1)
