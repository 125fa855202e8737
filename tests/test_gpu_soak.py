"""Soak of the hand-rolled ordering: the same batches run a few hundred times, every frame of every pass compared with the
first pass.  What it leans on: octave_tail_kernel's flag hand-off between workgroups (with its time-out and re-run), the
fence-free cross-stream events (hipEventDisableSystemFence), the counter block of the NEXT image being reset by the min/max
pass of the current one, the record blocks the groups of an image reserve from different streams, sixteen lanes sharing
the device.  A once-in-ten-thousand ordering bug shows up here before it shows up on an eight-GPU run."""
import numpy as np
import pytest

from util import kp_multiset_digest, smooth_noise, white_noise, assert_same_keypoints

pytestmark = pytest.mark.gpu


def _soak(bp, dev, passes, what):
    first = None
    for it in range(passes):
        got = bp.keypoints_batch(dev)
        dig = [kp_multiset_digest(g) for g in got]
        if first is None:
            first, keep = dig, got
            assert all(d[0] > 50 for d in dig)
        else:
            bad = [i for i, (a, b) in enumerate(zip(dig, first)) if a != b]
            assert not bad, "%s: pass %d, frames %r differ from the first pass (%r vs %r)" % (what, it, bad, [dig[i][0] for i in bad], [first[i][0] for i in bad])
    return keep


def test_c4_batch_200_passes(siftlib, oracle):
    """16 frames of 2048 x 2048 through the default 16-lane BatchPlan (the C4 shape), 200 times."""
    import torch
    import sift_pyocl_amd as sp
    shape = (2048, 2048)
    frames = [white_noise(shape, seed=2000 + i) if i % 4 else smooth_noise(shape, seed=2000 + i, sigma=2.5) for i in range(16)]
    dev = [torch.from_numpy(f).cuda() for f in frames]
    bp = sp.BatchPlan(shape=shape, dtype=np.float32)
    assert bp.lanes == 16
    got = _soak(bp, dev, 200, "c4 shape, 16 lanes")
    assert_same_keypoints(got[3], oracle.keypoints(frames[3]), "soaked batch, frame 3 against the oracle")


def test_two_lanes_of_4096_100_passes(siftlib):
    """Four 4096 x 4096 frames through two multi-stream lanes (three prioritised streams each), 100 times -- and the same
    frames through a single plan, call after call, which alternates the two counter blocks 400 times."""
    import torch
    import sift_pyocl_amd as sp
    shape = (4096, 4096)
    frames = [white_noise(shape, seed=3000 + i) for i in range(4)]
    dev = [torch.from_numpy(f).cuda() for f in frames]
    bp = sp.BatchPlan(shape=shape, dtype=np.float32, octave_max=3)
    assert bp.lanes == 2
    got = _soak(bp, dev, 100, "4096^2, 2 lanes")
    want = [kp_multiset_digest(g) for g in got]
    plan = sp.SiftPlan(shape=shape, dtype=np.float32, octave_max=3)
    for it in range(100):
        for i, d in enumerate(dev):
            assert kp_multiset_digest(plan.keypoints(d)) == want[i], "single plan, pass %d frame %d" % (it, i)


def test_small_frames_all_octaves_300_passes(siftlib):
    """512 x 512 frames with every octave: the forked later-octave chains, the tail launch and three record blocks per image,
    300 batches of 8 frames on 8 single-stream lanes and 300 calls of a three-stream plan."""
    import torch
    import sift_pyocl_amd as sp
    shape = (512, 512)
    frames = [smooth_noise(shape, seed=4000 + i, sigma=1.5 + 0.25 * i) for i in range(8)]
    dev = [torch.from_numpy(f).cuda() for f in frames]
    bp = sp.BatchPlan(shape=shape, dtype=np.float32)
    got = _soak(bp, dev, 300, "512^2, %d lanes" % bp.lanes)
    want = [kp_multiset_digest(g) for g in got]
    plan = sp.SiftPlan(shape=shape, dtype=np.float32)
    for it in range(300):
        i = it % 8
        assert kp_multiset_digest(plan.keypoints(dev[i])) == want[i], "single plan, call %d" % it
