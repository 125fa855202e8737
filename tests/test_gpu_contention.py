"""The two inter-workgroup wait protocols under REAL contention (not fault injection): octave_tail_kernel chains its workgroups
through a flag with a bounded wait and a host re-run, descriptor_open has every workgroup of a descriptor launch poll a word
that workgroup 0 publishes -- both lean on workgroups of one launch being resident together.  Here a second PROCESS keeps the
same GPU full of 16-lane batches of 2048 x 2048 frames while this one runs 512 x 512 frames with every octave (forked later
chains, the tail launch, three record blocks per image) and 1024 x 1024 frames through a three-stream plan.  Asserted: every
result equals the first one (a time-out costs a re-run, never a result); reported and bounded: how often the tail timed out."""
import os
import subprocess
import sys
import tempfile
import time

import numpy as np
import pytest

from util import kp_multiset_digest, smooth_noise, assert_same_keypoints

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_wait_protocols_beside_a_foreign_process(siftlib, oracle):
    import torch
    import sift_pyocl_amd as sp
    flag = tempfile.mktemp(prefix="siftmi_contention_")
    open(flag, "w").close()
    worker = subprocess.Popen([sys.executable, os.path.join(HERE, "contention_worker.py"), flag, "240"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    try:
        small = [smooth_noise((512, 512), seed=5000 + i, sigma=1.5 + 0.25 * i) for i in range(6)]
        mid = [smooth_noise((1024, 1024), seed=5100 + i, sigma=2.0) for i in range(2)]
        dsmall = [torch.from_numpy(f).cuda() for f in small]
        dmid = [torch.from_numpy(f).cuda() for f in mid]
        # the results to hold, taken while the GPU is still ours
        p_small = sp.SiftPlan(shape=(512, 512), dtype=np.float32)
        p_mid = sp.SiftPlan(shape=(1024, 1024), dtype=np.float32)
        bp = sp.BatchPlan(shape=(512, 512), dtype=np.float32)
        want_small = [kp_multiset_digest(p_small.keypoints(d)) for d in dsmall]
        want_mid = [kp_multiset_digest(p_mid.keypoints(d)) for d in dmid]
        assert kp_multiset_digest(bp.keypoints_batch(dsmall)[2]) == want_small[2]
        assert_same_keypoints(p_small.keypoints(dsmall[0]), oracle.keypoints(small[0]), "512^2 frame against the oracle")
        assert p_small.tail_timeouts() == (0, True)
        t0 = time.time()
        while not os.path.exists(flag + ".ready"):
            assert worker.poll() is None, "the contention worker died: %s" % worker.stderr.read()[-2000:]
            assert time.time() - t0 < 180, "the contention worker never got going"
            time.sleep(0.2)
        calls = 0
        t0 = time.time()
        while calls < 1500 and time.time() - t0 < 60:
            i = calls % 6
            assert kp_multiset_digest(p_small.keypoints(dsmall[i])) == want_small[i], "512^2 call %d beside the foreign process" % calls
            if calls % 5 == 0:
                assert kp_multiset_digest(p_mid.keypoints(dmid[calls % 2])) == want_mid[calls % 2], "1024^2 call %d" % calls
            if calls % 25 == 0:
                got = bp.keypoints_batch(dsmall)
                assert [kp_multiset_digest(g) for g in got] == want_small, "512^2 batch at call %d" % calls
            calls += 1
        assert calls >= 200, "only %d calls in a minute: the GPU was not shared, it was starved" % calls
    finally:
        if os.path.exists(flag):
            os.remove(flag)
        try:
            out, err = worker.communicate(timeout=120)
        except subprocess.TimeoutExpired:
            worker.kill()
            out, err = worker.communicate()
        if os.path.exists(flag + ".ready"):
            os.remove(flag + ".ready")
    line = [l for l in out.splitlines() if l.startswith("CONTENTION_WORKER")]
    assert line, "no report from the contention worker: %s" % err[-2000:]
    fields = dict(kv.split("=", 1) for kv in line[0].split()[1:3])
    assert int(fields["batches"]) >= 20 and int(fields["mismatches"]) == 0, line[0]       # it really ran beside us, and stayed right itself
    # the bounded waits: a plan that timed out once has dropped the one-launch form (at most ONE re-run per plan, ever)
    n_small, on_small = p_small.tail_timeouts()
    n_batch, lanes_on = bp.tail_timeouts()
    print("contention: %d calls beside %s batches of the foreign process; tail time-outs: single plan %d (one-launch form %s), "
          "batch lanes %d (%d of %d lanes keep it)" % (calls, fields["batches"], n_small, "kept" if on_small else "dropped", n_batch, lanes_on, bp.lanes))
    assert n_small <= 1 and n_batch <= bp.lanes
