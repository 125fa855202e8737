"""Worker of tests/test_gpu_multi.py: one rank per GPU under torch.distributed.run, backend "nccl" (RCCL over xGMI).

Checks, on every rank, that the sharded paths return what ONE rank computes on its own:
  * batch.keypoints_batch(frames): frame i on rank i mod N through a BatchPlan, all-gather of counts + records on
    device tensors (gather_records_device) -- against a single-rank BatchPlan over all frames, byte for byte per frame;
  * the same with fewer frames than ranks (an empty shard joins the collectives with nothing);
  * batch.match_sharded(kp1, kp2): queries split over the ranks, MatchPlan per rank -- against MatchPlan.match of the whole
    lists on one device (as sets of pairs; the reference defines no order).
Exit code 0 = every rank agreed.  Not collected by pytest (no test_ prefix)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = int(os.environ.get("LOCAL_RANK", rank))
    # SIFT_MULTI_REHEARSAL=1: every rank on cuda:0 with the collectives over gloo (RCCL refuses two ranks on one device) --
    # the one-GPU rehearsal of this very script; the exchange then takes the host-staged path of batch.keypoints_batch
    rehearsal = os.environ.get("SIFT_MULTI_REHEARSAL") == "1"
    if rehearsal:
        local = 0
        os.environ["LOCAL_RANK"] = "0"                      # the plans pick their device from it
        torch.cuda.set_device(0)
        dist.init_process_group(backend="gloo")
    else:
        assert torch.cuda.device_count() >= world, "one GPU per rank"
        torch.cuda.set_device(local)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
    backend = dist.get_backend()
    assert dist.get_world_size() == world and backend == ("gloo" if rehearsal else "nccl")
    import sift_pyocl_amd as sp
    from sift_pyocl_amd import batch
    from util import sort_kp, sort_rows

    size, n_frames = 1024, 2 * world + 1                  # an uneven split: rank 0 owns one frame more
    frames = [torch.from_numpy(np.random.default_rng(100 + i).random((size, size), dtype=np.float32)).cuda() for i in range(n_frames)]
    got = batch.keypoints_batch(frames)
    single = sp.BatchPlan(shape=(size, size), dtype=np.float32, device=local)
    exp = single.keypoints_batch(frames)
    assert len(got) == n_frames
    for i in range(n_frames):
        assert len(got[i]) == len(exp[i]) > 100, (rank, i, len(got[i]), len(exp[i]))
        assert sort_kp(got[i]).tobytes() == sort_kp(exp[i]).tobytes(), "frame %d differs on rank %d" % (i, rank)
    # fewer frames than ranks: the last rank owns nothing
    few = batch.keypoints_batch(frames[:world - 1]) if world > 1 else []
    for i in range(world - 1):
        assert sort_kp(few[i]).tobytes() == sort_kp(exp[i]).tobytes()
    # MatchPlan sharded by query
    a, b = exp[0], exp[1]
    b2 = b.copy(); b2[:len(a) // 2] = a[:len(a) // 2]       # half of the second list are exact copies: known matches
    pairs = batch.match_sharded(a, b2)
    whole = sp.MatchPlan(size=max(len(a), len(b2)), device=local).match(a, b2, raw_results=True)
    assert len(pairs) == len(whole) >= len(a) // 4
    assert np.array_equal(sort_rows(np.asarray(pairs)), sort_rows(np.asarray(whole)))
    dist.barrier()
    if rank == 0:
        print("multi ok: %d ranks over %s, %d frames, %d pairs" % (world, backend, n_frames, len(pairs)), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
