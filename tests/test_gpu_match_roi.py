"""GPU parity for SURVEY 8f-3: ROI-masked matching (`matching_valid`, matching_cpu.cl:136-199) and the mutual-best
extension, through the C ABI (siftmi_match_set_roi / siftmi_match_ex) against the oracle (itself pinned against the
reference's kernel built natively, tests/test_oracle_vs_ref.py)."""
import numpy as np
import pytest

from util import dtype_kp, smooth_noise, sort_rows

pytestmark = pytest.mark.gpu


def lists(n1, n2, shared, seed, H=90, W=120, spill=True):
    rng = np.random.default_rng(seed)
    a = np.zeros(n1, dtype_kp); b = np.zeros(n2, dtype_kp)
    a["desc"] = rng.integers(0, 256, (n1, 128), dtype=np.uint8)
    b["desc"] = rng.integers(0, 256, (n2, 128), dtype=np.uint8)
    idx = rng.permutation(n1)[:shared]
    b["desc"][:shared] = np.clip(a["desc"][idx].astype(int) + rng.integers(-6, 7, (shared, 128)), 0, 255).astype(np.uint8)
    b["desc"][shared + 1] = b["desc"][shared]                       # an exact duplicate: tie-breaking matters
    ext = 1.15 if spill else 0.999
    a["x"] = rng.random(n1) * W * ext; a["y"] = rng.random(n1) * H * ext
    b["x"] = rng.random(n2) * W * ext; b["y"] = rng.random(n2) * H * ext
    roi = (rng.random((H, W)) > 0.3).astype(np.int8)
    return a, b, roi


def check(mp, oracle, a, b, roi, mode, mutual, what):
    got = mp.match(a, b, raw_results=True, roi_mode=mode, mutual=mutual)
    want, n = oracle.match_ex(a, b, roi, mode, mutual=mutual, cap=max(1, len(a)))
    assert len(got) == n, "%s: %d pairs, oracle %d" % (what, len(got), n)
    assert np.array_equal(sort_rows(got), sort_rows(want)), what
    return n


@pytest.mark.parametrize("n1,n2", [(700, 650), (3000, 5000), (257, 64), (5, 3000)])
def test_strict_and_mutual_vs_oracle(siftlib, oracle, n1, n2):
    import sift_pyocl_amd as sp
    a, b, roi = lists(n1, n2, min(n1, n2) // 2, seed=n1 + n2)
    mp = sp.MatchPlan()
    mp.set_roi(roi)
    plain = check(mp, oracle, a, b, None, 0, False, "roi set but roi_mode 0 (reference behaviour: ignored)")
    ref_plain = mp.match(a, b, raw_results=True)
    assert len(ref_plain) == plain
    n_strict = check(mp, oracle, a, b, roi, 2, False, "strict")
    n_mut = check(mp, oracle, a, b, None, 0, True, "mutual")
    n_both = check(mp, oracle, a, b, roi, 2, True, "strict + mutual")
    assert n_strict <= plain and n_mut <= plain and n_both <= n_strict
    if min(n1, n2) >= 64:
        assert n_strict > 0 and n_mut > 0


def test_matching_valid_literal(siftlib, oracle):
    """roi_mode 1 = the reference kernel's literal semantics, including its quirks."""
    import sift_pyocl_amd as sp
    a, b, roi = lists(900, 800, 500, seed=4, spill=False)
    mp = sp.MatchPlan()
    mp.set_roi(roi)
    # (1) every list-2 keypoint on a valid pixel: only the list-1 drop rule acts
    ys, xs = np.nonzero(roi)
    rng = np.random.default_rng(8)
    pick = rng.integers(0, len(ys), len(b))
    b["x"] = xs[pick] + rng.random(len(b)).astype(np.float32) * 0.9; b["y"] = ys[pick] + rng.random(len(b)).astype(np.float32) * 0.9
    n = check(mp, oracle, a, b, roi, 1, False, "literal, list 2 all valid")
    assert 0 < n < 500
    # list-1 keypoints beyond the mask array are processed, not dropped
    a2 = a.copy(); a2["x"] += 500.0
    n2 = check(mp, oracle, a2, b, roi, 1, False, "literal, list 1 outside the array")
    assert n2 >= n
    # (2) exactly one masked-out list-2 keypoint: distance 0 to everything -> every query pairs with it
    b2 = b.copy(); zy, zx = np.argwhere(roi == 0)[0]
    b2["x"][17] = zx + 0.5; b2["y"][17] = zy + 0.25
    got = mp.match(a, b2, raw_results=True, roi_mode="reference")
    want, nw = oracle.match_ex(a, b2, roi, 1, cap=len(a))
    assert len(got) == nw and np.array_equal(sort_rows(got), sort_rows(want)) and (got[:, 1] == 17).all()
    # (3) two masked-out list-2 keypoints: dist1 == dist2 == 0 -> no pair at all
    b2["x"][99] = zx + 0.1; b2["y"][99] = zy + 0.1
    assert len(mp.match(a, b2, raw_results=True, roi_mode=1)) == 0 == oracle.match_ex(a, b2, roi, 1)[1]
    check(mp, oracle, a, b2, roi, 1, True, "literal + mutual")
    # unset_roi: roi_mode needs a mask
    mp.unset_roi()
    with pytest.raises(RuntimeError):
        mp.match(a, b, roi_mode=1)


def test_mutual_on_real_keypoints(siftlib, oracle):
    import sift_pyocl_amd as sp
    from util import smooth_noise
    big = smooth_noise((700, 760), seed=21, sigma=2.0)
    i1 = np.ascontiguousarray(big[10:650, 20:724]); i2 = np.ascontiguousarray(big[17:657, 9:713])
    plan = sp.SiftPlan(template=i1)
    k1 = plan.keypoints(i1); k2 = plan.keypoints(i2)
    mp = sp.MatchPlan()
    roi = np.zeros(i1.shape, np.int8); roi[100:500, 150:600] = 1
    mp.set_roi(roi)
    for mode, mutual in ((0, True), (2, False), (2, True), (1, False)):
        check(mp, oracle, k1, k2, roi, mode, mutual, "real keypoints mode %s mutual %s" % (mode, mutual))
    rec = mp.match(k1, k2, roi_mode="strict", mutual=True)
    assert rec.shape[1] == 2 and len(rec) > 50
    # strict: both ends of every pair are on the mask
    assert roi[rec[:, 0].y.astype(int), rec[:, 0].x.astype(int)].all() and roi[rec[:, 1].y.astype(int), rec[:, 1].x.astype(int)].all()
    assert abs(np.median(rec[:, 1].x - rec[:, 0].x) - 11.0) < 0.1 and abs(np.median(rec[:, 1].y - rec[:, 0].y) + 7.0) < 0.1


def test_query_slices_equal_the_full_scan(siftlib):
    """The multi-GPU form of MatchPlan (batch.match_sharded) splits the queries over the ranks: on the HIP path the union
    of the slices' pairs (first index shifted by the slice start) must be the single-device result, and match_sharded
    without a process group is that single-device call."""
    import sift_pyocl_amd as sp
    from sift_pyocl_amd.batch import match_sharded
    img = smooth_noise((600, 700), seed=21)
    plan = sp.SiftPlan(template=img)
    a = plan.keypoints(img)
    b = plan.keypoints(img + smooth_noise((600, 700), seed=22) * 0.03)
    mp = sp.MatchPlan()
    full = mp.match(a, b, raw_results=True)
    assert len(full) > 50
    parts = []
    for r in range(3):
        lo, hi = r * len(a) // 3, (r + 1) * len(a) // 3
        part = mp.match(a[lo:hi], b, raw_results=True).copy()
        part[:, 0] += lo
        parts.append(part)
    assert np.array_equal(sort_rows(np.concatenate(parts)), sort_rows(full))
    assert np.array_equal(sort_rows(match_sharded(a, b, plan=mp)), sort_rows(full))
