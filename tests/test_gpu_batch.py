"""GPU parity for the batched, pipelined path: every frame of a batch must come back bit-identical to the
oracle / to SiftPlan.keypoints of the same frame, whatever the lane count, input residency or frame order."""
import numpy as np
import pytest

from util import assert_same_keypoints, smooth_noise, white_noise

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("lanes", [1, 3, 4])
def test_batch_equals_oracle_per_frame(siftlib, oracle, lanes):
    import sift_pyocl_amd as sp
    shape = (300, 421)
    frames = [smooth_noise(shape, seed=40 + i, sigma=1.5 + 0.3 * (i % 3)) if i % 2 else white_noise(shape, seed=40 + i) for i in range(7)]
    bp = sp.BatchPlan(shape=shape, dtype=np.float32, lanes=lanes)
    got = bp.keypoints_batch(frames)
    assert len(got) == len(frames)
    for i, (g, f) in enumerate(zip(got, frames)):
        assert_same_keypoints(g, oracle.keypoints(f), "frame %d, %d lanes" % (i, lanes))
    # second batch on the same plan, different order and length: no state leaks between frames or batches
    got2 = bp.keypoints_batch(frames[::-1][:5])
    for g, f in zip(got2, frames[::-1][:5]):
        assert_same_keypoints(g, oracle.keypoints(f), "second batch")
    assert bp.keypoints_batch([]) == []
    assert_same_keypoints(bp.keypoints(frames[2]), oracle.keypoints(frames[2]), "batch of one")


def test_batch_device_frames_and_typed_frames(siftlib, oracle):
    import torch
    import sift_pyocl_amd as sp
    shape = (1024, 1100)
    frames = [(smooth_noise(shape, seed=60 + i) * 60000 / 1.0).clip(0, 65535).astype(np.uint16) for i in range(5)]
    want = [oracle.keypoints(f.astype(np.float32)) for f in frames]
    bp = sp.BatchPlan(template=frames[0], lanes=2)
    for g, w in zip(bp.keypoints_batch(frames), want):
        assert_same_keypoints(g, w, "uint16 host frames")
    dev = [torch.from_numpy(f.astype(np.int32)).to(torch.int32).cuda() for f in frames]
    bp32 = sp.BatchPlan(shape=shape, dtype=np.int32, lanes=3)
    for g, w in zip(bp32.keypoints_batch(dev), want):
        assert_same_keypoints(g, w, "int32 device frames")
    with pytest.raises(RuntimeError):
        bp32.keypoints_batch([dev[0], frames[1].astype(np.int32)])          # mixed residency


def test_batch_2048_matches_single_plan(siftlib):
    """BASELINE.json configs[3] shape: 2048 x 2048 frames; the batch must equal the frame-by-frame plan bit for bit."""
    import sift_pyocl_amd as sp
    frames = [white_noise((2048, 2048), seed=80 + i) for i in range(8)]
    plan = sp.SiftPlan(shape=(2048, 2048), dtype=np.float32)
    bp = sp.BatchPlan(shape=(2048, 2048), dtype=np.float32, lanes=4)
    got = bp.keypoints_batch(frames)
    for i, f in enumerate(frames):
        assert_same_keypoints(got[i], plan.keypoints(f), "2048 frame %d" % i)
    from sift_pyocl_amd.batch import keypoints_batch
    again = keypoints_batch(frames[:3], lanes=2)
    for i in range(3):
        assert_same_keypoints(again[i], got[i], "keypoints_batch() helper")


@pytest.mark.parametrize("lanes", [1, 4])
def test_batch_of_blank_frames(siftlib, lanes):
    """A batch in which no frame has any keypoint (constant frames: the normalisation divides by zero, the reference does
    not guard either) must come back as empty arrays -- in parked mode (lanes > 2) nothing is parked at all."""
    import sift_pyocl_amd as sp
    shape = (200, 260)
    bp = sp.BatchPlan(shape=shape, dtype=np.float32, lanes=lanes)
    blank = np.zeros(shape, np.float32)
    got = bp.keypoints_batch([blank, blank + 3.0, blank])
    assert [len(g) for g in got] == [0, 0, 0]
    assert all(g.dtype == sp.SiftPlan.dtype_kp for g in got)
    assert len(bp.keypoints(blank)) == 0
    assert len(sp.SiftPlan(shape=shape, dtype=np.float32).keypoints(blank)) == 0
    # and a mixed batch still delivers the non-empty frame
    mixed = bp.keypoints_batch([blank, white_noise(shape, seed=3)])
    assert len(mixed[0]) == 0 and len(mixed[1]) > 10


def test_c4_shape_16_lanes_against_the_oracle(siftlib, oracle):
    """BASELINE.json configs[3] as bench.py --config c4 runs it on one rank: the default BatchPlan for 2048 x 2048 frames
    (16 parked lanes), 16 device-resident frames.  A sample of the frames is compared with the ORACLE (not with SiftPlan),
    and the device-side hand-back (keypoints_batch_device -> split_gathered, the single-rank form of the exchange)
    must deliver the same records as keypoints_batch."""
    import torch
    import sift_pyocl_amd as sp
    from sift_pyocl_amd.batch import split_gathered
    shape = (2048, 2048)
    frames = [white_noise(shape, seed=1000 + i) if i % 4 else smooth_noise(shape, seed=1000 + i, sigma=2.0) for i in range(16)]
    dev = [torch.from_numpy(f).cuda() for f in frames]
    bp = sp.BatchPlan(shape=shape, dtype=np.float32)            # default lanes: what the bench uses
    assert bp.lanes == 16
    got = bp.keypoints_batch(dev)
    assert len(got) == 16
    for i in (0, 5, 10, 15):
        assert_same_keypoints(got[i], oracle.keypoints(frames[i]), "c4 frame %d, 16 lanes" % i)
    counts, records = bp.keypoints_batch_device(dev)
    assert counts == [len(g) for g in got]
    assert records.numel() == sum(counts) * 144
    back = split_gathered([counts], records.view(1, -1), 16, 1)
    for i in range(16):
        assert_same_keypoints(back[i], got[i], "device hand-back, frame %d" % i)


def test_tail_timeout_reruns_the_frame(siftlib, oracle):
    """a time-out of octave_tail_kernel must re-run the frame -- for a single plan (siftmi_plan_keypoints) and
    for a lane of a batch (batch_retire), where it used to fail the whole call.  Option "tail_fault" = n treats the next n
    tail launches as timed out; the results must not change, and afterwards the plan walks the small octaves launch by launch."""
    import sift_pyocl_amd as sp
    frames = [white_noise((256, 320), seed=40 + i) for i in range(6)]
    want = [oracle.keypoints(f) for f in frames]
    plan = sp.SiftPlan(template=frames[0])
    plan.set_option("tail_fault", 1)
    for f, w in zip(frames[:3], want[:3]):              # the first call is the one that is run twice
        assert_same_keypoints(plan.keypoints(f), w, "single plan, injected tail time-out")
    bp = sp.BatchPlan(template=frames[0], lanes=3)
    bp.set_option("tail_fault", 1)                      # every lane re-runs its first frame
    for got, w in zip(bp.keypoints_batch(frames), want):
        assert_same_keypoints(got, w, "batch lane, injected tail time-out")
    for got, w in zip(bp.keypoints_batch(frames), want):
        assert_same_keypoints(got, w, "batch, call after the re-run")
