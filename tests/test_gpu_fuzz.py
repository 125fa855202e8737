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
