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


def test_descriptor_kernel_forms_agree(siftlib, oracle):
    """The row-interval descriptor kernel (default), the streaming form forced through the option, and plans with the
    widest windows the 64-tap limit of the blur schedule allows (init_sigma = 4.0: up to 243 window rows, still inside the
    row tables) must all equal the oracle bit for bit."""
    import sift_pyocl_amd as sp
    img = smooth_noise((420, 510), seed=23, sigma=2.5)
    want = oracle.keypoints(img)
    plan = sp.SiftPlan(template=img)
    assert_same_keypoints(plan.keypoints(img), want, "row-interval form")
    plan.set_option("desc_stream", 1)
    assert_same_keypoints(plan.keypoints(img), want, "streaming form")
    plan.set_option("desc_stream", 0)
    # one launch holds a wave-per-keypoint and a workgroup-per-keypoint form; the group's count picks one ("desc_team")
    plan.set_option("desc_team", 1 << 30)
    assert_same_keypoints(plan.keypoints(img), want, "workgroup-per-keypoint form for every group")
    plan.set_option("desc_team", 0)
    assert_same_keypoints(plan.keypoints(img), want, "wave-per-keypoint form for every group")
    plan.set_option("desc_team", 1024)
    # the orientation launch likewise ("ori_team")
    plan.set_option("ori_team", 1 << 30)
    assert_same_keypoints(plan.keypoints(img), want, "orientation: workgroup per keypoint for every group")
    plan.set_option("ori_team", 0)
    assert_same_keypoints(plan.keypoints(img), want, "orientation: wave per keypoint for every group")
    plan.set_option("ori_team", 1024)
    # full gradient maps instead of the lazy gradient ("maps": 1 always, 0 never, 2 by the previous image), every form
    plan.set_option("maps", 1)
    assert_same_keypoints(plan.keypoints(img), want, "gradient maps")
    for team in (1 << 30, 0):
        plan.set_option("desc_team", team)
        plan.set_option("ori_team", team)
        assert_same_keypoints(plan.keypoints(img), want, "gradient maps, team option %d" % team)
    plan.set_option("desc_team", 1024)
    plan.set_option("ori_team", 1024)
    plan.set_option("overlap", 0)
    assert_same_keypoints(plan.keypoints(img), want, "gradient maps, one stream")
    plan.set_option("overlap", 1)
    plan.set_option("maps", 2)
    # round 6: the marching blur's priority feedback ("march_prio": never / by rule / every launch) does not touch a result
    for v in (0, 2, 1):
        plan.set_option("march_prio", v)
        assert_same_keypoints(plan.keypoints(img), want, "march_prio %d" % v)
    with pytest.raises(RuntimeError):
        plan.set_option("march_prio", 3)
    plan.pinned_results = False                                  # plain numpy result + device-to-host copy
    assert_same_keypoints(plan.keypoints(img), want, "unpinned result array")
    for init_sigma in (3.0, 4.0):
        big = sp.SiftPlan(template=img, init_sigma=init_sigma)
        got = big.keypoints(img)
        assert len(got) > 10
        assert_same_keypoints(got, oracle.keypoints(img, par=oracle.default_params(init_sigma=init_sigma)), "init_sigma %g" % init_sigma)
        big.set_option("desc_stream", 1)
        assert_same_keypoints(big.keypoints(img), got, "init_sigma %g, streaming form" % init_sigma)
        big.set_option("desc_stream", 0)
        big.set_option("desc_team", 1 << 30)
        assert_same_keypoints(big.keypoints(img), got, "init_sigma %g, workgroup-per-keypoint form" % init_sigma)


@pytest.mark.parametrize("shape", [(26, 26), (22, 40), (40, 21), (31, 64)])
def test_clipped_windows_in_both_descriptor_forms(siftlib, oracle, shape):
    """Frames barely larger than a descriptor window: every window is clipped by the frame, many hold fewer than 256
    samples -- in the workgroup-per-keypoint form some waves then never get a batch, and their (never written) routing
    tables must read as empty.  Both forms against the oracle, a small detection border so that keypoints sit at the edge;
    several fresh plans, because what a never-written table holds is whatever the previous kernel left in LDS."""
    import sift_pyocl_amd as sp
    from sift_pyocl_amd.param import par
    saved = dict(par)
    try:
        par.update(dict(BorderDist=2))
        opar = oracle.default_params()
        opar.border_dist = 2
        found = 0
        for seed in range(4):
            img = smooth_noise(shape, seed=300 + seed, sigma=1.2)
            want = oracle.keypoints(img, par=opar)
            found += len(want)
            for team in (1 << 30, 0):
                for maps in (0, 1):
                    plan = sp.SiftPlan(template=img)
                    plan.set_option("desc_team", team)
                    plan.set_option("ori_team", team)
                    plan.set_option("maps", maps)
                    assert_same_keypoints(plan.keypoints(img), want, "%r seed %d desc_team=%d maps=%d" % (shape, seed, team, maps))
        assert found > 0
    finally:
        par.update(saved)


def test_result_arrays_outlive_the_plan_and_recycle(siftlib):
    """Results are views of pinned blocks from the library's pool: they must stay valid after later calls and after the
    plan is gone, and dropping them must hand the blocks back (no growth over many calls)."""
    import gc
    import sift_pyocl_amd as sp
    img = smooth_noise((256, 300), seed=5)
    plan = sp.SiftPlan(template=img)
    first = plan.keypoints(img)
    keep = first.copy()
    others = [plan.keypoints(np.roll(img, s, axis=1)) for s in range(1, 6)]
    assert np.array_equal(first.view(np.uint8), keep.view(np.uint8))      # not overwritten by the later calls
    del plan, others
    gc.collect()
    assert np.array_equal(first.view(np.uint8), keep.view(np.uint8))


SMALL_SHAPES = [(512, 512), (300, 517), (129, 1000), (97, 97), (260, 131), (640, 480), (1030, 770)]


@pytest.mark.parametrize("shape", SMALL_SHAPES)
def test_small_frame_kernels_agree(siftlib, oracle, shape):
    """Small planes take their own launch shapes -- 32 x 16 blur tiles, short extrema strips, and one launch for all
    octaves of at most 64 x 64 samples (octave_tail_kernel, a workgroup per octave chained through plane 3).  Every
    combination of those options must give the oracle's records bit for bit; odd sizes give the tail planes odd pitches."""
    import sift_pyocl_amd as sp
    img = smooth_noise(shape, seed=shape[0] + shape[1], sigma=2.0)
    want = oracle.keypoints(img) if shape[0] * shape[1] <= 300 * 517 else None
    plan = sp.SiftPlan(template=img)
    base = plan.keypoints(img)
    assert len(base) > 5
    if want is not None:
        assert_same_keypoints(base, want, "defaults vs oracle %r" % (shape,))
    for opts in (dict(tail=0), dict(tail=0, ext_rows=32), dict(tail=1, ext_rows=8), dict(tail=1, overlap=0),
                 dict(desc_team=0), dict(desc_team=1 << 30, fork=0), dict(fork=1), dict(early_chain=0), dict(early_chain=1, fork=0),
                 dict(desc_team=0, desc_dynamic=0, desc_blocks=333), dict(desc_team=0, desc_blocks=7, ori_blocks=77), dict(ori_team=0), dict(ori_team=1 << 30, ori_blocks=5), dict(fused_shrink=0), dict(fused_shrink=1, overlap=0, tail=0), dict(fused_refine=0), dict(fused_refine=2, tail=0),
                 dict(xcd_map=0), dict(split=1, fork=0), dict(split=1, fork=0, early_chain=0, tail=0), dict(split=1, fork=0, xcd_map=0, fused_refine=0)):
        other = sp.SiftPlan(template=img)
        for name, value in opts.items():
            other.set_option(name, value)
        assert_same_keypoints(other.keypoints(img), base, "%r with %r" % (shape, opts))


def test_double_im_size(siftlib, oracle):
    """par.DoubleImSize: the reference only counts the input as blurred by sigma 1.0 instead of 0.5 (plan.py:254, 297, 534:
    the initial blur becomes sqrt(1.6^2 - 1) wide, 11 taps; nothing is resampled).  Read by the constructor (the taps it
    prepares) and again at call time (the sigma it looks up): a value changed in between is the reference's KeyError."""
    import math
    import sift_pyocl_amd as sp
    from sift_pyocl_amd.param import par
    saved = dict(par)
    img = smooth_noise((333, 402), seed=17, sigma=2.0)
    img16 = (img / img.max() * 65535).astype(np.uint16)
    try:
        plain = sp.SiftPlan(template=img).keypoints(img)
        par["DoubleImSize"] = 1
        plan = sp.SiftPlan(template=img)
        s0, n0 = plan.gaussian_sizes()[0]
        assert n0 == 11 and abs(s0 - math.sqrt(1.6 ** 2 - 1.0)) < 1e-12
        want = oracle.keypoints(img, par=oracle.default_params(double_im_size=1))
        got = plan.keypoints(img)
        assert_same_keypoints(got, want, "DoubleImSize = 1")
        assert len(got) != len(plain)                        # the parameter really changed the result
        assert_same_keypoints(plan.keypoints(img), want, "DoubleImSize = 1, second call")
        # a typed frame (no fused form of an 11-tap initial blur: the convert pass) and the batched path
        want16 = oracle.keypoints(img16.astype(np.float32), par=oracle.default_params(double_im_size=1))
        assert_same_keypoints(sp.SiftPlan(template=img16).keypoints(img16), want16, "DoubleImSize = 1, uint16")
        bp = sp.BatchPlan(template=img, lanes=2)
        for r in bp.keypoints_batch([img, img, img]):
            assert_same_keypoints(r, want, "DoubleImSize = 1, BatchPlan")
        # init_sigma <= 1.0: no initial blur at all with the parameter set (plan.py:535)
        flat = sp.SiftPlan(template=img, init_sigma=0.9)
        assert len(flat.gaussian_sizes()) == 5
        assert_same_keypoints(flat.keypoints(img), oracle.keypoints(img, par=oracle.default_params(init_sigma=0.9, double_im_size=1)),
                              "DoubleImSize = 1, init_sigma 0.9")
        # changed after the constructor: the reference finds no taps for the sigma it now computes (plan.py:585)
        par["DoubleImSize"] = 0
        with pytest.raises(KeyError):
            plan.keypoints(img)
        with pytest.raises(KeyError):
            bp.keypoints_batch([img])
        # ... unless no initial blur is needed either way
        assert_same_keypoints(sp.SiftPlan(template=img, init_sigma=0.4).keypoints(img),
                              oracle.keypoints(img, par=oracle.default_params(init_sigma=0.4)), "init_sigma 0.4")
        par["DoubleImSize"] = 1
        assert_same_keypoints(plan.keypoints(img), want, "DoubleImSize back to 1")
    finally:
        par.update(saved)
