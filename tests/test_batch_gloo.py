"""CPU test of the multi-GPU exchange step with the gloo backend, world_size 2: image sharding
(round robin) and the all-gather of keypoint records.  The per-rank keypoints come from the CPU
oracle here (the HIP path needs a GPU); the collective logic is what is under test."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, n_items, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ["SIFTMI_STANDALONE"] = "1"
    import torch.distributed as dist
    from oracle import pyoracle
    from sift_pyocl_amd.batch import gather_records, shard_indices
    from util import smooth_noise
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = shard_indices(n_items, rank, world)
    local = [pyoracle.keypoints(smooth_noise((96 + 8 * i, 120), seed=100 + i)) for i in mine]
    allk = gather_records(local, n_items, rank, world)
    q.put((rank, [k.tobytes() for k in allk]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_items", [5, 2, 1])     # 1: rank 1 owns no frame
def test_gather_records_world2(n_items):
    import torch.multiprocessing as mp
    from oracle import pyoracle
    from util import smooth_noise
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000 + n_items
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_items, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=240) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    expected = [pyoracle.keypoints(smooth_noise((96 + 8 * i, 120), seed=100 + i)).tobytes() for i in range(n_items)]
    assert results[0] == expected and results[1] == expected


def _worker_device_form(rank, world, port, n_items, q):
    """gather_records_device + split_gathered (the RCCL path's code, here on CPU tensors over gloo), incl. an empty shard"""
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ["SIFTMI_STANDALONE"] = "1"
    import torch
    import torch.distributed as dist
    from oracle import pyoracle
    from sift_pyocl_amd.batch import gather_records_device, shard_indices, split_gathered
    from util import smooth_noise
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = shard_indices(n_items, rank, world)
    local = [pyoracle.keypoints(smooth_noise((96 + 8 * i, 120), seed=100 + i)) for i in mine]
    counts = [len(k) for k in local]
    raw = b"".join(np.ascontiguousarray(k).tobytes() for k in local)
    records = torch.frombuffer(bytearray(raw), dtype=torch.uint8) if raw else torch.empty(0, dtype=torch.uint8)
    all_counts, gathered = gather_records_device(counts, records, n_items, rank, world)
    allk = split_gathered(all_counts, gathered, n_items, world)
    q.put((rank, [k.tobytes() for k in allk]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_items", [3, 1])
def test_gather_records_device_form_world2(n_items):
    import torch.multiprocessing as mp
    from oracle import pyoracle
    from util import smooth_noise
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000 + n_items
    procs = [ctx.Process(target=_worker_device_form, args=(r, 2, port, n_items, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=240) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    expected = [pyoracle.keypoints(smooth_noise((96 + 8 * i, 120), seed=100 + i)).tobytes() for i in range(n_items)]
    assert results[0] == expected and results[1] == expected


def _worker_match(rank, world, port, q):
    """match_sharded over gloo: queries split over two ranks, the CPU oracle as the matcher"""
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ["SIFTMI_STANDALONE"] = "1"
    import torch.distributed as dist
    from oracle import pyoracle
    from sift_pyocl_amd.batch import match_sharded
    from util import smooth_noise
    dist.init_process_group("gloo", rank=rank, world_size=world)
    a = pyoracle.keypoints(smooth_noise((150, 180), seed=7))
    b = pyoracle.keypoints(smooth_noise((150, 180), seed=7) + smooth_noise((150, 180), seed=8) * 0.05)
    pairs = match_sharded(a, b, matcher=lambda x, y: pyoracle.match(x, y)[0])
    q.put((rank, pairs.tobytes(), len(pairs)))
    dist.barrier()
    dist.destroy_process_group()


def test_match_sharded_world2():
    import torch.multiprocessing as mp
    from oracle import pyoracle
    from util import smooth_noise, sort_rows
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker_match, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    a = pyoracle.keypoints(smooth_noise((150, 180), seed=7))
    b = pyoracle.keypoints(smooth_noise((150, 180), seed=7) + smooth_noise((150, 180), seed=8) * 0.05)
    want, total = pyoracle.match(a, b)
    assert total > 20
    for _, raw, n in results:
        got = np.frombuffer(raw, dtype=np.int32).reshape(n, 2)
        assert np.array_equal(sort_rows(got), sort_rows(want))


def test_shard_indices():
    from sift_pyocl_amd.batch import shard_indices
    assert shard_indices(64, 3, 8) == list(range(3, 64, 8))
    assert sorted(sum((shard_indices(10, r, 4) for r in range(4)), [])) == list(range(10))
