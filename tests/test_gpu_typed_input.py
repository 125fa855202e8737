"""GPU parity for typed frames: integer / RGB inputs are converted at the point of use by the min/max
and the initial blur (preprocess.cl:53-223 fused away).  The result must be bit-identical to the keypoints of the
explicitly converted float32 frame (the reference's u8_to_float / rgb_to_float ... followed by the f32 pipeline),
which in turn is compared with the oracle."""
import os

import numpy as np
import pytest

from util import assert_same_keypoints, smooth_noise

pytestmark = pytest.mark.gpu


def frame(dtype, shape, seed):
    f = smooth_noise(shape, seed=seed, sigma=2.0)
    f = (f - f.min()) / (f.max() - f.min())
    info = np.iinfo(dtype)
    # wide types: use large magnitudes so that (float)x actually rounds (more than 24 significant bits)
    span = float(min(info.max, 2 ** 62))
    lo = 0.0 if info.min == 0 else -span / 2
    return (lo + f.astype(np.float64) * span * 0.99).astype(dtype)


def as_f32(img):
    if img.ndim == 3:
        r, g, b = (img[..., c].astype(np.float32) for c in range(3))
        return (np.float32(0.299) * r + np.float32(0.587) * g) + np.float32(0.114) * b
    return img.astype(np.float32)          # numpy int -> float32 rounds to nearest even, as the OpenCL cast


@pytest.mark.parametrize("shape", [(200, 300), (600, 1100)])          # tile kernel / marching kernel for the initial blur
@pytest.mark.parametrize("name", ["uint8", "uint16", "uint32", "uint64", "int32", "int64"])
def test_integer_frames(siftlib, oracle, name, shape):
    import sift_pyocl_amd as sp
    img = frame(np.dtype(name), shape, seed=len(name) + shape[0])
    ref32 = as_f32(img)
    want = oracle.keypoints(ref32)
    plan = sp.SiftPlan(template=img)
    got = plan.keypoints(img)
    assert len(got) > 50
    assert_same_keypoints(got, want, "%s frame %s" % (name, shape))
    mn, mx = plan.minmax()
    assert mn == ref32.min() and mx == ref32.max()
    # the convert-pass fallback gives the same bits
    plain = sp.SiftPlan(template=img)
    plain.set_option("fused_convert", 0)
    assert_same_keypoints(plain.keypoints(img), want, "%s frame, convert pass" % name)


@pytest.mark.parametrize("shape", [(180, 260), (520, 1030)])
def test_rgb_frames(siftlib, oracle, shape):
    import sift_pyocl_amd as sp
    rng = np.random.default_rng(shape[1])
    base = frame(np.uint8, shape, 5)
    rgb = np.stack([base, np.roll(base, 3, axis=1), 255 - base], axis=-1)
    rgb = np.clip(rgb.astype(int) + rng.integers(-3, 4, rgb.shape), 0, 255).astype(np.uint8)
    want = oracle.keypoints(as_f32(rgb))
    got = sp.SiftPlan(template=rgb).keypoints(rgb)
    assert len(got) > 50
    assert_same_keypoints(got, want, "RGB frame %s" % (shape,))


def test_typed_frame_without_initial_blur_and_other_sigma(siftlib, oracle):
    """init_sigma <= 0.5 -> plain normalise reads the typed frame; init_sigma = 2.0 -> 17-tap initial blur, for which no
    typed instantiation exists -> convert pass.  Both must equal the oracle on the converted frame."""
    import sift_pyocl_amd as sp
    img = frame(np.uint16, (300, 400), 9)
    for sig in (0.5, 2.0):
        par = oracle.default_params(init_sigma=sig)
        want = oracle.keypoints(as_f32(img), par)
        got = sp.SiftPlan(template=img, init_sigma=sig).keypoints(img)
        assert_same_keypoints(got, want, "uint16, init_sigma %s" % sig)


def test_device_resident_typed_frame(siftlib, oracle):
    import torch
    import sift_pyocl_amd as sp
    img = frame(np.uint8, (1024, 1280), 11)
    want = oracle.keypoints(as_f32(img))
    plan = sp.SiftPlan(template=img)
    t = torch.from_numpy(img).cuda()
    assert_same_keypoints(plan.keypoints(t), want, "device-resident uint8 frame")
    # a deliberately misaligned device view (offset by one row + 1 byte is not possible for a 2-D contiguous view;
    # use a fresh tensor sliced from a larger flat buffer at an odd offset) takes the convert-pass fallback
    flat = torch.empty(img.size + 64, dtype=torch.uint8, device="cuda")
    view = flat[3:3 + img.size].view(img.shape)
    view.copy_(t)
    assert view.data_ptr() % 16 != 0
    assert_same_keypoints(plan.keypoints(view), want, "misaligned device uint8 frame")
