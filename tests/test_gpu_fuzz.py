"""Randomised HIP-vs-oracle parity over odd shapes, input types and image statistics (a short, seeded version of
tools/dev/fuzz_parity.py): exercises the edges of the marching / tile blur kernels, the typed-frame paths and BatchPlan on
shapes no hand-written case lists.  Every frame must be bit-identical to the oracle."""
import numpy as np
import pytest

from util import assert_same_keypoints, smooth_noise, white_noise

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [11, 12])
def test_fuzz_shapes_and_dtypes(siftlib, oracle, seed):
    import sift_pyocl_amd as sp
    rng = np.random.default_rng(seed)
    for it in range(8):
        H = int(rng.integers(300, 1700)); W = int(rng.integers(700, 2300))
        if it % 4 == 0:
            H, W = W, H                                   # tall frames: W may drop below 1024 -> tile kernel
        kind = it % 3
        img = white_noise((H, W), seed=100 * seed + it) if kind == 0 else smooth_noise((H, W), seed=100 * seed + it, sigma=1.0 + (it % 4))
        dt = [np.float32, np.uint8, np.uint16, np.float32][it % 4]
        if dt != np.float32:
            img = ((img - img.min()) / (img.max() - img.min()) * np.iinfo(dt).max).astype(dt)
        want = oracle.keypoints(img.astype(np.float32))
        got = sp.SiftPlan(template=img).keypoints(img)
        assert_same_keypoints(got, want, "fuzz %d/%d %dx%d %s" % (seed, it, H, W, np.dtype(dt).name))
        if it % 5 == 0:
            for g in sp.BatchPlan(template=img, lanes=2).keypoints_batch([img, img]):
                assert_same_keypoints(g, want, "fuzz batch %d/%d" % (seed, it))


def test_fuzz_small_shapes(siftlib, oracle):
    """Small odd frames (down to two octaves): 32 x 16 blur tiles on planes that are not multiples of anything, the tail
    kernel with odd LDS pitches, the workgroup-per-keypoint forms of the orientation / descriptor launches, the fused
    hand-off and refinement -- with the stream layouts alternated (a short version of tools/dev/fuzz_small.py)."""
    import sift_pyocl_amd as sp
    rng = np.random.default_rng(21)
    for it in range(10):
        H = int(rng.integers(40, 420)); W = int(rng.integers(40, 560))
        img = white_noise((H, W), seed=700 + it) if it % 3 == 0 else smooth_noise((H, W), seed=700 + it, sigma=1.0 + (it % 4))
        dt = [np.float32, np.uint8, np.uint16, np.float32][it % 4]
        if dt != np.float32:
            img = ((img - img.min()) / (img.max() - img.min()) * np.iinfo(dt).max).astype(dt)
        want = oracle.keypoints(img.astype(np.float32))
        plan = sp.SiftPlan(template=img)
        if it % 4 == 1:
            plan.set_option("desc_team", 0); plan.set_option("ori_team", 0)
        if it % 4 == 2:
            plan.set_option("overlap", 0)
        if it % 4 == 3:
            plan.set_option("fork", 0); plan.set_option("early_chain", 0)
        assert_same_keypoints(plan.keypoints(img), want, "small fuzz %d %dx%d %s" % (it, H, W, np.dtype(dt).name))
        assert_same_keypoints(plan.keypoints(img), want, "small fuzz %d, second call" % it)
