"""Worker of tests/test_gpu_multi.py::test_rccl_call_path_on_one_gpu: ONE rank, backend "nccl" (RCCL), on the one GPU a
lease has.  It cannot show two ranks talking; it does show everything else of the N > 1 path before an 8-GPU node does:
RCCL is loaded and builds a communicator in the process that also holds libsiftmi.so's HIP runtime (the hazard
_lib._share_hip_runtime_with_torch exists for), and the device-tensor collectives of the exchange -- all_gather_into_tensor
of the count table and of the padded record bytes (gather_records_device), of pair counts and pairs (gather_pairs) -- run on
tensors the descriptor kernels wrote, with split_gathered taking them apart again.  Not collected by pytest."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    import sift_pyocl_amd as sp                                    # libsiftmi.so first, RCCL after it: one HIP runtime for both
    from sift_pyocl_amd import batch
    from util import sort_kp, sort_rows
    size, n = 1024, 5
    frames = [torch.from_numpy(np.random.default_rng(300 + i).random((size, size), dtype=np.float32)).cuda() for i in range(n)]
    bp = sp.BatchPlan(shape=(size, size), dtype=np.float32, device=0)
    want = bp.keypoints_batch(frames)
    dist.init_process_group(backend="nccl", device_id=torch.device("cuda", 0))
    assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
    # the exchange of the batched path, on device tensors end to end
    counts, records = bp.keypoints_batch_device(frames)
    assert records.is_cuda and records.numel() == 144 * sum(counts)
    all_counts, gathered = batch.gather_records_device(counts, records, n, 0, 1)
    assert gathered.is_cuda and all_counts == [counts]
    back = batch.split_gathered(all_counts, gathered, n, 1)
    for i in range(n):
        assert len(back[i]) == len(want[i]) > 100
        assert sort_kp(back[i]).tobytes() == sort_kp(want[i]).tobytes(), "frame %d" % i
    # ... of the plan-level entry point with a forced exchange (world size 1 normally skips it)
    again = batch.keypoints_batch(frames, plan=bp)
    assert all(sort_kp(a).tobytes() == sort_kp(w).tobytes() for a, w in zip(again, want))
    # ... and of the sharded match
    a, b = want[0], want[1].copy()
    b[:len(a) // 2] = a[:len(a) // 2]
    whole = sp.MatchPlan(size=max(len(a), len(b)), device=0).match(a, b, raw_results=True)
    pairs = batch.gather_pairs(np.asarray(whole, dtype=np.int32), 1)
    assert np.array_equal(sort_rows(pairs), sort_rows(np.asarray(whole))) and len(pairs) >= len(a) // 4
    # a SiftPlan keeps working after the communicator exists (both users of the HIP runtime alive in one process)
    plan = sp.SiftPlan(shape=(size, size), dtype=np.float32, device=0)
    assert sort_kp(plan.keypoints(frames[2])).tobytes() == sort_kp(want[2]).tobytes()
    t = torch.ones(4, device="cuda")
    dist.all_reduce(t)
    assert float(t.sum().item()) == 4.0
    dist.barrier()
    print("rccl ws1 ok: %d frames, %d record bytes gathered, %d pairs" % (n, int(gathered.numel()), len(pairs)), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
