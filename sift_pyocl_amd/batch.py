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
    (``siftmi_batch_*`` in include/siftmi.h).  Same constructor keywords as ``SiftPlan`` plus ``lanes`` (default: 16 for frames up to 2048 x 2048, 8 below 600 x 600, 2 for larger ones);
    ``keypoints_batch(images)`` returns one recarray per frame, each bit-identical to ``SiftPlan.keypoints``.
    """

    def __init__(self, *args, **kwargs):
        lanes = kwargs.pop("lanes", None)
        if lanes is None:
            # measured on MI355X: many one-stream lanes for small frames (dependent-launch latency), two three-stream lanes
            # for large ones (one frame nearly fills the GPU)
            shape = kwargs.get("shape") or (kwargs["template"].shape if kwargs.get("template") is not None else (args[0] if args else None))
            # (64 frames, ms per batch, 8 / 16 / 32 lanes: 2048^2 21.6 / 20.0 / 18.9, 1024^2 14.7 / 12.9, 512^2 10.2 / 11.1 --
            # the host thread needs ~0.12 ms to enqueue a frame, which is what bounds the smallest frames)
            px = int(shape[0]) * int(shape[1]) if shape is not None else 0
            lanes = 2 if px == 0 or px > 2048 * 2048 else (16 if px > 600 * 600 else 8)
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

    def set_option(self, name, value):
        """SiftPlan.set_option on every lane (``siftmi_batch_set_option``); results never depend on an option."""
        _lib.check(_lib.lib().siftmi_batch_set_option(self._handle, str(name).encode(), int(value)))

    def tail_timeouts(self):
        """(re-runs after a time-out of the small octaves' one-launch form, lanes that still use it): see SiftPlan.tail_timeouts"""
        n, on = C.c_int64(), C.c_int32()
        _lib.check(_lib.lib().siftmi_batch_tail_timeouts(self._handle, C.byref(n), C.byref(on)))
        return int(n.value), int(on.value)

    def blur_times(self):
        """profile='light': hipEvent time, launches and pixels of the full-resolution blur launches of the last batch"""
        ms = C.c_double(); nl = C.c_int64(); px = C.c_double()
        _lib.check(_lib.lib().siftmi_batch_blur_ms(self._handle, C.byref(ms), C.byref(nl), C.byref(px)))
        return {"blur0_ms": ms.value, "blur0_launches": nl.value, "blur0_pixels": px.value}

    def _sync_params(self, L):
        """`par` is read at call time, as in the reference"""
        key = (par.PeakThresh, par.EdgeThresh1, par.EdgeThresh, par.OriSigma, par.BorderDist, par.DoubleImSize)
        if key != self._par_key:
            params = self._current_params()
            _lib.check(L.siftmi_batch_set_params(self._handle, C.byref(params)))
            self._params, self._par_key = params, key

    def _marshal(self, images):
        """pointer table, element-type code, residency flag and keep-alive list of a batch of frames"""
        n = len(images)
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
        return ptrs, code, dev_flags.pop(), keep

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
            self._sync_params(L)
            ptrs, code, is_dev, keep = self._marshal(images)
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
            _lib.check(L.siftmi_batch_keypoints_into(self._handle, ptrs, n, code, is_dev, outs if direct else None,
                                                     caps if direct else None, counts, offsets, C.byref(parked), C.byref(ovf)))
            self.overflow = bool(ovf.value)
            if self.overflow:
                logger.warning("Keypoint counter overflow: an octave of a frame needs more than %s entries, result cut to that per octave", self.kpsize)
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

    def keypoints_batch_device(self, images):
        """Same as ``keypoints_batch`` but the records stay in HBM: returns ``(counts, records)`` where `records` is a
        torch uint8 tensor on this plan's device holding the 144-byte records of frame 0, frame 1, ... back to back
        (``sum(counts) * 144`` bytes).  This is what the multi-GPU exchange (``gather_records_device``) consumes: no
        host staging between the descriptor kernels and the RCCL all-gather."""
        import torch
        images = list(images)
        n = len(images)
        dev = torch.device("cuda", self.device)
        if n == 0:
            return [], torch.empty(0, dtype=torch.uint8, device=dev)
        with self._sem:
            L = _lib.lib()
            self._sync_params(L)
            ptrs, code, is_dev, keep = self._marshal(images)
            counts = (C.c_int64 * n)()
            offsets = (C.c_int64 * n)()
            parked = C.c_int64(0)
            ovf = C.c_int32(0)
            _lib.check(L.siftmi_batch_keypoints_into(self._handle, ptrs, n, code, is_dev, None, None, counts, offsets,
                                                     C.byref(parked), C.byref(ovf)))
            self.overflow = bool(ovf.value)
            out = torch.empty(max(1, parked.value) * RECORD_BYTES, dtype=torch.uint8, device=dev)
            if parked.value:
                _lib.check(L.siftmi_batch_fetch(self._handle, out.data_ptr(), 1, 0, parked.value))
            # the arena holds the frames in retirement order; hand them back in input order
            cnt = [int(c) for c in counts]
            off = [int(o) for o in offsets]
            if any(off[i] != sum(cnt[:i]) for i in range(n)):
                parts = [out[off[i] * RECORD_BYTES:(off[i] + cnt[i]) * RECORD_BYTES] for i in range(n)]
                out = torch.cat(parts) if parts else out
            self._records_per_frame = max(1.0, sum(cnt) / float(n))
            del keep
        return cnt, out[:sum(cnt) * RECORD_BYTES]

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


def gather_records_device(counts, records, n_items, rank, world_size, group=None):
    """The exchange step on device tensors (backend "nccl" = RCCL over xGMI): all-gather of the per-image counts, then
    of the record bytes padded to the largest per-rank total.  Nothing is staged through the host; the only host
    read-back is the (world_size x per_rank) count table that sizes the payload.

    :param counts: records per owned image, in the order of shard_indices(n_items, rank, world_size)
    :param records: torch uint8 device tensor, the owned images' records back to back
    :return: (all_counts, gathered): all_counts[r][j] = records of the j-th image of rank r (python ints); gathered =
             device uint8 tensor (world_size, max_total * 144): row r holds rank r's records back to back
    """
    import torch
    import torch.distributed as dist

    dev = records.device
    per_rank = (n_items + world_size - 1) // world_size
    mine = torch.zeros(per_rank, dtype=torch.int64)
    mine[:len(counts)] = torch.tensor([int(c) for c in counts], dtype=torch.int64) if counts else mine[:0]
    mine = mine.to(dev)
    table = torch.empty(world_size * per_rank, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(table, mine, group=group)
    table = table.view(world_size, per_rank)
    all_counts = table.cpu().tolist()                       # 8 bytes per image: sizes the payload
    max_total = max(1, max(sum(row) for row in all_counts))
    payload = torch.zeros(max_total * RECORD_BYTES, dtype=torch.uint8, device=dev)
    payload[:records.numel()] = records
    gathered = torch.empty(world_size * payload.numel(), dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(gathered, payload, group=group)
    return all_counts, gathered.view(world_size, -1)


def split_gathered(all_counts, gathered, n_items, world_size):
    """Per-image numpy recarrays from the result of gather_records_device (one device-to-host copy of everything)."""
    host = gathered.cpu().numpy()
    out = [None] * n_items
    for r in range(world_size):
        at = 0
        for j, idx in enumerate(shard_indices(n_items, r, world_size)):
            n = int(all_counts[r][j])
            out[idx] = host[r, at:at + n * RECORD_BYTES].copy().view(SiftPlan.dtype_kp).view(numpy.recarray)
            at += n * RECORD_BYTES
    return out


def keypoints_batch(images, plan=None, rank=None, world_size=None, gather=True, device=None, **plan_kwargs):
    """Keypoints of a list of same-shape images, sharded over the ranks of the default process group.

    Each rank runs its share through a ``BatchPlan`` (pipelined, one result copy); without an initialised process
    group everything runs on one GPU.  Every rank must pass the same `images` list (only the owned ones are touched).
    A ``SiftPlan`` passed as `plan` is used frame by frame.  With the "nccl" backend (RCCL) the exchange runs on device
    tensors end to end (``gather_records_device``); other backends (the gloo rehearsal on CPU) stage through the host.
    """
    import torch.distributed as dist

    if rank is None or world_size is None:
        if dist.is_available() and dist.is_initialized():
            rank, world_size = dist.get_rank(), dist.get_world_size()
        else:
            rank, world_size = 0, 1
    mine = shard_indices(len(images), rank, world_size)
    plan_given = plan is not None
    if plan is None and mine:
        plan = BatchPlan(template=images[mine[0]], **plan_kwargs)
    # The exchange path must be the same on every rank, so it is chosen from rank-independent facts only: the backend
    # and the kind of plan the caller passed (a rank that owns no frame has no plan at all: plan is None there).
    nccl = gather and world_size > 1 and dist.is_initialized() and dist.get_backend() == "nccl"
    on_device = nccl and (plan is None or isinstance(plan, BatchPlan))
    if nccl:
        # ... and agreed explicitly: a rank without frames (plan is None) would pick the device path while ranks that were
        # handed a frame-by-frame SiftPlan pick the host-staged one -- two different collective sequences, i.e. a hang; the
        # same if only some ranks pass a plan.  One 12-byte all-reduce (MIN) settles "every rank can take the device path"
        # for everybody and shows whether `plan` was given on all ranks or on none (MIN of the flag and of its negation);
        # every rank takes part whatever it was handed, so a mismatch is an error on every rank instead of a hang on some.
        # (Cost: one small collective + one .item() per batch, ~30 us beside milliseconds of frames.)
        import torch
        flag = torch.tensor([1 if on_device else 0, 1 if plan_given else 0, 0 if plan_given else 1], dtype=torch.int32,
                            device=torch.device("cuda", torch.cuda.current_device()))
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        agreed = [int(v) for v in flag.tolist()]
        if agreed[1] == 0 and agreed[2] == 0:
            raise RuntimeError("keypoints_batch: `plan` was passed on some ranks and not on others; pass it on every rank or on none")
        on_device = bool(agreed[0])
    if on_device:
        import torch
        if isinstance(plan, BatchPlan):
            counts, records = plan.keypoints_batch_device([images[i] for i in mine])
        else:                                # empty shard (fewer frames than ranks): joins the collectives with nothing
            counts = []
            records = torch.empty(0, dtype=torch.uint8, device=torch.device("cuda", torch.cuda.current_device()))
        all_counts, gathered = gather_records_device(counts, records, len(images), rank, world_size)
        return split_gathered(all_counts, gathered, len(images), world_size)
    if nccl and device is None:              # frame-by-frame SiftPlan under RCCL: the collectives need device tensors
        import torch
        device = torch.device("cuda", torch.cuda.current_device())
    if isinstance(plan, BatchPlan):
        local = plan.keypoints_batch([images[i] for i in mine])
    else:
        local = [plan.keypoints(images[i]) for i in mine] if mine else []
    if not gather or world_size == 1:
        if world_size == 1:
            return local
        return {i: k for i, k in zip(mine, local)}
    return gather_records(local, len(images), rank, world_size, device=device)


def match_sharded(kp1, kp2, plan=None, rank=None, world_size=None, device=None, matcher=None):
    """``MatchPlan.match(kp1, kp2, raw_results=True)`` with the QUERIES split over the ranks: rank r scans the
    contiguous slice ``kp1[r * n1 // N : (r + 1) * n1 // N]`` against the whole second list (every rank holds it: 14.4 MB
    at 100 k keypoints), then one all-gather of the pair counts and one of the (i, j) pairs -- a query's result does not
    depend on any other query (matching_cpu.cl:67-108), so the union is the single-device result (in rank order; the
    reference does not define an order either).  Every rank returns all the pairs.

    :param kp1, kp2: host recarrays of 144-byte records, the same on every rank
    :param plan: a ``MatchPlan`` of this rank (created when None); `matcher(a, b) -> (n, 2) int array` replaces it (tests)
    """
    import torch
    import torch.distributed as dist

    if rank is None or world_size is None:
        if dist.is_available() and dist.is_initialized():
            rank, world_size = dist.get_rank(), dist.get_world_size()
        else:
            rank, world_size = 0, 1
    n1 = len(kp1)
    lo, hi = rank * n1 // world_size, (rank + 1) * n1 // world_size
    if matcher is None:
        if plan is None:
            from .match import MatchPlan
            plan = MatchPlan(size=max(1, min(hi - lo, len(kp2))))
        matcher = lambda a, b: plan.match(a, b, raw_results=True)      # noqa: E731
    mine = numpy.asarray(matcher(kp1[lo:hi], kp2), dtype=numpy.int32).reshape(-1, 2) if hi > lo and len(kp2) else numpy.empty((0, 2), numpy.int32)
    mine = mine.copy()
    mine[:, 0] += lo
    if world_size == 1:
        return mine
    return gather_pairs(mine, world_size, device=device)


def gather_pairs(mine, world_size, device=None, group=None):
    """The exchange of ``match_sharded``: all-gather of the pair counts, then of the (i, j) pairs padded to the largest
    count; with the "nccl" backend on device tensors.  Every rank returns the pairs of all ranks, in rank order."""
    import torch
    import torch.distributed as dist

    nccl = dist.get_backend(group) == "nccl"
    dev = torch.device(device) if device is not None else (torch.device("cuda", torch.cuda.current_device()) if nccl else torch.device("cpu"))
    counts = torch.empty(world_size, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(counts, torch.tensor([len(mine)], dtype=torch.int64, device=dev), group=group)
    counts = counts.cpu().tolist()
    width = max(1, max(counts))
    payload = torch.zeros((width, 2), dtype=torch.int32)
    payload[:len(mine)] = torch.from_numpy(numpy.ascontiguousarray(mine, dtype=numpy.int32).reshape(-1, 2))
    gathered = torch.empty((world_size * width, 2), dtype=torch.int32, device=dev)
    dist.all_gather_into_tensor(gathered, payload.to(dev), group=group)
    gathered = gathered.cpu().numpy().reshape(world_size, width, 2)
    return numpy.concatenate([gathered[r, :counts[r]] for r in range(world_size)], axis=0)
