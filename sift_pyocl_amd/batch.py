"""Batched multi-GPU path: independent images sharded over the GPUs of one node, one process per GPU
(torch.distributed; backend "nccl" is RCCL over xGMI on ROCm, "gloo" on CPU for tests), and ONE exchange
step at the end: an all-gather of the per-image keypoint records.

The reference has no multi-device code at all (one pyopencl.Context per plan, sift-src/plan.py:183);
a user would build one plan per device.  Images are independent, so the data path needs no
collective; only the result hand-back is a collective, and it is small (144 B per keypoint).
"""
import numpy

from .plan import SiftPlan

RECORD_BYTES = 144


def shard_indices(n_items, rank, world_size):
    """Round-robin ownership: image i is processed by rank i % world_size."""
    return list(range(rank, n_items, world_size))


def gather_records(local_records, n_items, rank, world_size, device=None, group=None):
    """All-gather per-image record arrays.

    :param local_records: list of (n_i,) dtype_kp arrays for the images owned by this rank, in the
                          order of shard_indices(n_items, rank, world_size)
    :return: list of n_items arrays (every rank gets every image's keypoints)
    """
    import torch
    import torch.distributed as dist

    mine = shard_indices(n_items, rank, world_size)
    assert len(mine) == len(local_records)
    per_rank = (n_items + world_size - 1) // world_size
    counts = torch.zeros(per_rank, dtype=torch.int64)
    for j, rec in enumerate(local_records):
        counts[j] = len(rec)
    dev = torch.device(device) if device is not None else torch.device("cpu")
    counts = counts.to(dev)
    all_counts = torch.empty(world_size * per_rank, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(all_counts, counts, group=group)
    all_counts = all_counts.cpu().view(world_size, per_rank)
    # one padded byte buffer per rank; the payload is latency- not bandwidth-bound
    max_total = int(all_counts.sum(dim=1).max().item())
    payload = torch.zeros(max(1, max_total) * RECORD_BYTES, dtype=torch.uint8)
    at = 0
    for rec in local_records:
        raw = numpy.ascontiguousarray(rec).view(numpy.uint8).reshape(-1)
        payload[at:at + raw.size] = torch.from_numpy(raw.copy())
        at += raw.size
    payload = payload.to(dev)
    gathered = torch.empty(world_size * payload.numel(), dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(gathered, payload, group=group)
    gathered = gathered.cpu().numpy().reshape(world_size, -1)
    out = [None] * n_items
    for r in range(world_size):
        at = 0
        for j, idx in enumerate(shard_indices(n_items, r, world_size)):
            n = int(all_counts[r, j].item())
            chunk = gathered[r, at:at + n * RECORD_BYTES]
            out[idx] = chunk.copy().view(SiftPlan.dtype_kp).view(numpy.recarray)
            at += n * RECORD_BYTES
    return out


def keypoints_batch(images, plan=None, rank=None, world_size=None, gather=True, device=None, **plan_kwargs):
    """Keypoints of a list of same-shape images, sharded over the ranks of the default process group.

    Without an initialised process group this degenerates to a plain loop on one GPU.
    Every rank must pass the same `images` list (only the owned ones are touched).
    """
    import torch.distributed as dist

    if rank is None or world_size is None:
        if dist.is_available() and dist.is_initialized():
            rank, world_size = dist.get_rank(), dist.get_world_size()
        else:
            rank, world_size = 0, 1
    mine = shard_indices(len(images), rank, world_size)
    if plan is None and mine:
        plan = SiftPlan(template=images[mine[0]], **plan_kwargs)
    local = [plan.keypoints(images[i]) for i in mine]
    if not gather or world_size == 1:
        if world_size == 1:
            return local
        return {i: k for i, k in zip(mine, local)}
    return gather_records(local, len(images), rank, world_size, device=device)
