"""Second process of tests/test_gpu_contention.py: keeps the GPU busy with 16-lane batches of 2048 x 2048 frames (the C4
shape) until the file given as argv[1] disappears or argv[2] seconds have passed; prints how many batches it ran and whether
every pass returned what its first pass returned."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402


def main():
    flag, limit = sys.argv[1], float(sys.argv[2])
    import torch
    import sift_pyocl_amd as sp
    from util import kp_multiset_digest, white_noise
    shape = (2048, 2048)
    dev = [torch.from_numpy(white_noise(shape, seed=7000 + i)).cuda() for i in range(16)]
    bp = sp.BatchPlan(shape=shape, dtype=np.float32)
    first = [kp_multiset_digest(g) for g in bp.keypoints_batch(dev)]
    open(flag + ".ready", "w").close()              # the parent starts its own loop now
    t0 = time.time()
    batches, bad = 0, 0
    while os.path.exists(flag) and time.time() - t0 < limit:
        got = bp.keypoints_batch(dev)
        batches += 1
        if batches % 8 == 0:
            bad += [kp_multiset_digest(g) for g in got] != first
    print("CONTENTION_WORKER batches=%d mismatches=%d tail=%r" % (batches, bad, bp.tail_timeouts()), flush=True)


if __name__ == "__main__":
    main()
