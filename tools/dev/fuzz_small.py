"""dev: randomized HIP-vs-oracle parity over small odd shapes (tile blur, tail kernel with odd pitches, team forms of the
orientation / descriptor launches, fused hand-off and refinement): python tools/dev/fuzz_small.py [seed] [cases]
(the CPU oracle is what takes the time: ~20 s per case on the GPU box; seed 1: 63 cases, seed 7: 70 cases (profiles/r05/fuzz_seed7.txt), all
bit-identical, two calls each)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import sift_pyocl_amd as sp
from oracle import pyoracle
from util import assert_same_keypoints, smooth_noise, white_noise
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
# optional: height range, width range (default: small odd shapes; "1000 2300 1024 2600" reaches the marching blur with odd pitches)
hlo, hhi, wlo, whi = (int(a) for a in sys.argv[3:7]) if len(sys.argv) > 6 else (40, 700, 40, 900)
t0 = time.time()
for it in range(n):
    H = int(rng.integers(hlo, hhi)); W = int(rng.integers(wlo, whi))
    kind = it % 3
    img = white_noise((H, W), seed=it) if kind == 0 else smooth_noise((H, W), seed=it, sigma=1.0 + (it % 4))
    dt = [np.float32, np.uint8, np.uint16, np.float32][it % 4]
    if dt != np.float32:
        img = ((img - img.min()) / (img.max() - img.min()) * np.iinfo(dt).max).astype(dt)
    want = pyoracle.keypoints(img.astype(np.float32))
    plan = sp.SiftPlan(template=img)
    if it % 4 == 1: plan.set_option("desc_team", 0); plan.set_option("ori_team", 0)
    if it % 4 == 2: plan.set_option("overlap", 0)
    if it % 4 == 3: plan.set_option("fork", 0)
    # (since the XCD-contiguous order: every eighth case builds the later octaves' pyramids on two streams into one group --
    # option "split" needs the unforked chain --, every fifth deals the tiles in plain workgroup order)
    if it % 8 == 7: plan.set_option("split", 1)
    if it % 5 == 4: plan.set_option("xcd_map", 0)
    # (round 6: the marching blur's priority feedback forced on every launch / off in a third of the cases each)
    if it % 3 == 1: plan.set_option("march_prio", 2)
    if it % 3 == 2: plan.set_option("march_prio", 0)
    got = plan.keypoints(img)
    assert_same_keypoints(got, want, "fuzz %d %dx%d %s" % (it, H, W, np.dtype(dt).name))
    assert_same_keypoints(plan.keypoints(img), want, "fuzz %d second call" % it)
    print("ok %2d  %4dx%4d %-7s %6d kp  octaves %d" % (it, H, W, np.dtype(dt).name, len(got), plan.octave_max), flush=True)
print("all %d cases bit-identical (%.0f s)" % (n, time.time() - t0))
