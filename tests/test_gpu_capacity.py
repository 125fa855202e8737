"""Keypoint capacity as the reference: kpsize = H*W // PIX_PER_KP entries PER OCTAVE (sift-src/plan.py:243; the buffer and its
counter are reset for every octave, plan.py:797-804, and each octave's results are read back on their own, plan.py:748-756).
An image within that rule returns every record -- up to octave_max * kpsize of them -- and one beyond it is flagged.  The
oracle (oracle/sift_oracle.c: so_keypoints) applies the rule per octave exactly as the reference's kernels do; the device
lists start at kpsize entries and grow on demand, so the first call of these plans also exercises the re-run path."""
import numpy as np
import pytest

from util import assert_same_keypoints, smooth_noise, sort_kp

pytestmark = pytest.mark.gpu


def _oracle_par(oracle, pix_per_kp):
    par = oracle.default_params()
    par.pix_per_kp = pix_per_kp
    return par


def test_total_beyond_kpsize_but_no_octave_overflows(siftlib, oracle):
    """512^2 smoothed noise: ~2400 keypoints, ~1950 of them in octave 0.  PIX_PER_KP = 120 gives kpsize 2184: every octave
    fits, the total does not -- the reference returns everything, and so must this build (round 4 cut at kpsize in total)."""
    import sift_pyocl_amd as sp
    img = smooth_noise((512, 512))
    want, ovf = oracle.keypoints(img, _oracle_par(oracle, 120), return_overflow=True)
    assert not ovf and len(want) > 2184
    plan = sp.SiftPlan(template=img, PIX_PER_KP=120)
    assert plan.kpsize == 2184
    got = plan.keypoints(img)
    assert not plan.overflow
    assert_same_keypoints(got, want, "PIX_PER_KP 120")
    rec, grows = plan.capacity()
    assert rec >= len(want) and grows >= 1                   # the record list started at kpsize and was grown under the first image
    again = plan.keypoints(img)                              # second call: nothing grows, same records
    assert_same_keypoints(again, want, "PIX_PER_KP 120, second call")
    assert plan.capacity() == (rec, grows)
    # the same through the un-pinned result path and the count-then-fetch path of large lists
    plan.pinned_results = False
    assert_same_keypoints(plan.keypoints(img), want, "PIX_PER_KP 120, plain result array")


@pytest.mark.parametrize("pix_per_kp", [60, 100, 140, 200, 260, 400, 1000])
def test_overflow_flag_follows_the_reference_rule(siftlib, oracle, pix_per_kp):
    """The flag is the reference's rule evaluated on the image's per-scale counts: equal to the oracle's for every capacity;
    without overflow the records are the oracle's, with it they are a subset of the uncapped result, at most kpsize per
    octave (which ones is as arbitrary as in the reference, whose atomic counter decides)."""
    import sift_pyocl_amd as sp
    img = smooth_noise((512, 512))
    want, ovf = oracle.keypoints(img, _oracle_par(oracle, pix_per_kp), return_overflow=True)
    plan = sp.SiftPlan(template=img, PIX_PER_KP=pix_per_kp)
    for call in range(2):
        got = plan.keypoints(img)
        assert plan.overflow == ovf, "PIX_PER_KP %d: overflow %r, oracle %r" % (pix_per_kp, plan.overflow, ovf)
        if not ovf:
            assert_same_keypoints(got, want, "PIX_PER_KP %d call %d" % (pix_per_kp, call))
        else:
            full = sort_kp(oracle.keypoints(img))
            assert len(got) <= plan.octave_max * plan.kpsize
            # (a multiset: two candidates that walk to the same sample give the same record twice, in the reference too)
            from collections import Counter
            have, all_of_them = Counter(r.tobytes() for r in got), Counter(r.tobytes() for r in full)
            assert not (have - all_of_them), "a record that the uncapped pipeline does not produce"
            assert len(got) >= len(want)                     # the oracle cuts inside the pipeline (later scales see a full buffer), never keeps more


def test_white_noise_capacities(siftlib, oracle):
    """White noise loses half of a scale's candidates in the refinement, so the candidate term of the rule (oriented keypoints
    of the octave so far + candidates of the scale > kpsize) is the one that fires first: the flag must still follow the oracle."""
    import sift_pyocl_amd as sp
    from util import white_noise
    img = white_noise((512, 512), seed=5)
    seen = set()
    for pix_per_kp in (1300, 1500, 1700, 2000, 2600):
        want, ovf = oracle.keypoints(img, _oracle_par(oracle, pix_per_kp), return_overflow=True)
        plan = sp.SiftPlan(template=img, PIX_PER_KP=pix_per_kp)
        got = plan.keypoints(img)
        assert plan.overflow == ovf, "PIX_PER_KP %d" % pix_per_kp
        if not ovf:
            assert_same_keypoints(got, want, "white PIX_PER_KP %d" % pix_per_kp)
        seen.add(ovf)
    assert seen == {False, True}


def test_batch_lanes_grow_their_lists(siftlib, oracle):
    """The lanes of a BatchPlan are plans of their own: each grows its lists under its first frame that needs it and runs
    that frame again; the batch's flag is the OR of its frames' (two of these five break the rule in octave 0)."""
    import sift_pyocl_amd as sp
    imgs = [smooth_noise((512, 512), seed=s) for s in (0, 1, 2, 3, 4)]
    for lanes in (1, 2):
        bp = sp.BatchPlan(template=imgs[0], PIX_PER_KP=120, lanes=lanes)
        for call in range(2):
            out = bp.keypoints_batch(imgs)
            flags = []
            for img, got in zip(imgs, out):
                want, ovf = oracle.keypoints(img, _oracle_par(oracle, 120), return_overflow=True)
                flags.append(ovf)
                if not ovf:
                    assert_same_keypoints(got, want, "batch PIX_PER_KP 120, %d lanes, call %d" % (lanes, call))
                else:
                    assert len(want) <= len(got) <= bp.octave_max * bp.kpsize
            assert bp.overflow == any(flags) and any(flags) and not all(flags)
