"""Batched multi-GPU path: independent images sharded over the GPUs of one node, one process per GPU
(torch.distributed; backend "nccl" is RCCL over xGMI on ROCm, "gloo" on CPU for tests), and ONE exchange
step at the end: an all-gather of the per-image keypoint records.

The reference has no multi-device code at all (one pyopencl.Context per plan, sift-src/plan.py:183);
a user would build one plan per device.  Images are independent, so the data path needs no
collective; only the result hand-back is a collective, and it is small (144 B per keypoint).
"""
import ctypes as C
import logging

import numpy

from . import _lib
from .param import par
from .plan import SiftPlan, _pointer_of

logger = logging.getLogger("sift.batch")
RECORD_BYTES = 144


class BatchPlan(SiftPlan):
    """Throughput form of ``SiftPlan`` for stacks of same-shape frames.

    ``lanes`` independent device plans take the frames round-robin; nothing waits on the host until a lane is
    reused, the records of the whole batch are parked on the device and come back in one copy
    (``siftmi_batch_*`` in include/siftmi.h).  Same constructor keywords as ``SiftPlan`` plus ``lanes`` (default: 8 for frames up to 2048 x 2048, else 2);
    ``keypoints_batch(images)`` returns one recarray per frame, each bit-identical to ``SiftPlan.keypoints``.
    """

    def __init__(self, *args, **kwargs):
        lanes = kwargs.pop("lanes", None)
        if lanes is None:
            # measured on MI355X: many one-stream lanes for small frames (dependent-launch latency), two three-stream lanes
            # for large ones (one frame nearly fills the GPU)
            shape = kwargs.get("shape") or (kwargs["template"].shape if kwargs.get("template") is not None else (args[0] if args else None))
            lanes = 8 if shape is not None and int(shape[0]) * int(shape[1]) <= 2048 * 2048 else 2
        self.lanes = int(lanes)
        self._records_per_frame = 4096.0
        self._light = kwargs.get("profile") == "light"
        if kwargs.get("profile") and not self._light:
            raise RuntimeError("BatchPlan only supports profile='light' (blur brackets); profile a SiftPlan for per-stage events")
        kwargs["profile"] = False
        SiftPlan.__init__(self, *args, **kwargs)

    def _create(self, L):
        _lib.check(L.siftmi_batch_create(self.shape[0], self.shape[1], self._code, self.device, C.byref(self._params),
                                         self.lanes, C.byref(self._handle)))
        nbytes = C.c_int64()
        _lib.check(L.siftmi_batch_info(self._handle, None, C.byref(nbytes)))
        self.memory = int(nbytes.value)
        if self._light:
            _lib.check(L.siftmi_batch_set_profile(self._handle, 1))

    def _destroy(self, L, h):
        L.siftmi_batch_destroy(h)

    def blur_times(self):
        """profile='light': hipEvent time, launches and pixels of the full-resolution blur launches of the last batch"""
        ms = C.c_double(); nl = C.c_int64(); px = C.c_double()
        _lib.check(_lib.lib().siftmi_batch_blur_ms(self._handle, C.byref(ms), C.byref(nl), C.byref(px)))
        return {"blur0_ms": ms.value, "blur0_launches": nl.value, "blur0_pixels": px.value}

    def keypoints_batch(self, images):
        """Keypoints of a sequence of frames (all numpy / host, or all device tensors).

        :return: list of numpy recarrays (x, y, scale, angle, desc[128]), one per frame, in input order
        """
        images = list(images)
        n = len(images)
        if n == 0:
            return []
        with self._sem:
            L = _lib.lib()
            key = (par.PeakThresh, par.EdgeThresh1, par.EdgeThresh, par.OriSigma, par.BorderDist, par.DoubleImSize)
            if key != self._par_key:
                params = self._current_params()
                _lib.check(L.siftmi_batch_set_params(self._handle, C.byref(params)))
                self._params, self._par_key = params, key
            ptrs = (C.c_void_p * n)()
            keep = []
            dev_flags = set()
            code = None
            for i, image in enumerate(images):
                ptr, is_dev, dtype, shape, k = _pointer_of(image)
                assert tuple(shape[:2]) == tuple(self.shape)
                assert dtype in [self.dtype, numpy.float32]
                if dtype == numpy.float32 and len(shape) == 2:
                    c = _lib.DTYPE_CODES["float32"]
                elif self.dtype == numpy.float64 and dtype == numpy.float64:
                    k = k.float() if is_dev else k.astype(numpy.float32)
                    ptr = k.data_ptr() if is_dev else k.ctypes.data
                    c = _lib.DTYPE_CODES["float32"]
                elif len(shape) == 3 and dtype == numpy.uint8 and self.RGB:
                    c = _lib.DTYPE_CODES["rgb8"]
                elif self.dtype in self.converter and len(shape) == 2:
                    c = _lib.DTYPE_CODES[self.dtype.name]
                else:
                    raise RuntimeError("invalid input format error (%s)" % (str(self.dtype)))
                if code is None:
                    code = c
                elif c != code:
                    raise RuntimeError("all frames of a batch must have the same element type")
                ptrs[i] = ptr
                keep.append(k)
                dev_flags.add(int(bool(is_dev)))
            if len(dev_flags) != 1:
                raise RuntimeError("the frames of a batch must be all host arrays or all device tensors")
            counts = (C.c_int64 * n)()
            offsets = (C.c_int64 * n)()
            parked = C.c_int64(0)
            ovf = C.c_int32(0)
            # Records are delivered frame by frame into host arrays while the batch runs.  Their capacity is a guess (1.5x
            # the records per frame of the previous batch); a frame that does not fit stays parked on the device and is
            # fetched afterwards.  Many lanes of small frames: the host must keep feeding the lanes, so everything is
            # parked and fetched with one copy at the end instead.
            direct = self.lanes <= 2
            outs = (C.c_void_p * n)()
            caps = (C.c_int64 * n)()
            arrays = [None] * n
            if direct:
                cap = int(1.5 * self._records_per_frame) + 256
                for i in range(n):
                    arrays[i] = numpy.empty(cap, dtype=self.dtype_kp)
                    outs[i] = arrays[i].ctypes.data
                    caps[i] = cap
            _lib.check(L.siftmi_batch_keypoints_into(self._handle, ptrs, n, code, dev_flags.pop(), outs if direct else None,
                                                     caps if direct else None, counts, offsets, C.byref(parked), C.byref(ovf)))
            self.overflow = bool(ovf.value)
            if self.overflow:
                logger.warning("Keypoint counter overflow: more than %s keypoints in a frame, result truncated", self.kpsize)
            flat = numpy.empty(0, dtype=self.dtype_kp)      # a batch of blank frames parks nothing
            if parked.value:
                flat = numpy.empty(parked.value, dtype=self.dtype_kp)
                _lib.check(L.siftmi_batch_fetch(self._handle, flat.ctypes.data, 0, 0, parked.value))
            result = []
            for i in range(n):
                if offsets[i] < 0:
                    result.append(arrays[i][:counts[i]].view(numpy.recarray))
                else:
                    result.append(flat[offsets[i]:offsets[i] + counts[i]].view(numpy.recarray))
            self._records_per_frame = max(1.0, sum(counts) / float(n))
            del keep
        return result

    def keypoints(self, image):
        return self.keypoints_batch([image])[0]

    __call__ = keypoints

    def minmax(self):
        raise RuntimeError("BatchPlan keeps no per-frame min/max; use SiftPlan")

    def kernel_times(self):
        raise RuntimeError("BatchPlan does not collect per-stage events; profile a SiftPlan instead")


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

    Each rank runs its share through a ``BatchPlan`` (pipelined, one result copy); without an initialised process
    group everything runs on one GPU.  Every rank must pass the same `images` list (only the owned ones are touched).
    A ``SiftPlan`` passed as `plan` is used frame by frame.
    """
    import torch.distributed as dist

    if rank is None or world_size is None:
        if dist.is_available() and dist.is_initialized():
            rank, world_size = dist.get_rank(), dist.get_world_size()
        else:
            rank, world_size = 0, 1
    mine = shard_indices(len(images), rank, world_size)
    if plan is None and mine:
        plan = BatchPlan(template=images[mine[0]], **plan_kwargs)
    if isinstance(plan, BatchPlan):
        local = plan.keypoints_batch([images[i] for i in mine])
    else:
        local = [plan.keypoints(images[i]) for i in mine]
    if not gather or world_size == 1:
        if world_size == 1:
            return local
        return {i: k for i, k in zip(mine, local)}
    return gather_records(local, len(images), rank, world_size, device=device)
