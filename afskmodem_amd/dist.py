"""Stream sharding across the GPUs of one node.

Streams are independent (nothing in afskmodem.py:354-381 couples two streams),
so the data path has no collective: rank r demodulates the contiguous range
``shard_range(n, r, world)`` with its own kernel launches.  The only exchange is
an optional gather of the decoded bytes + five int32 per stream, done as ONE
fixed-stride collective -- to one rank (``gather_flat_to_root``) or to all (``gather_flat``) --
over RCCL / xGMI with backend "nccl" (gloo on CPU tests).
"""
from __future__ import annotations

import numpy as np


def bind_to_device_numa_node(device=None):
    """One process per GPU belongs on the socket its GPU hangs off: confine the calling thread (and every thread
    it starts afterwards -- the library's I/O pool, torch's workers) to the CPUs of the NUMA node closest to
    ``device`` (default: the current one).  On the two-socket MI355X hosts a pinned buffer on the other socket is
    read across the inter-socket link (43 instead of 57 GB/s host -> device) and page-cache copies run at half the
    rate.  Returns {"node", "cpus"} or None when the node cannot be found out (then nothing is changed)."""
    import os
    import torch
    try:
        idx = torch.cuda.current_device() if device is None else torch.device(device).index or 0
        pr = torch.cuda.get_device_properties(idx)
        bdf = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        node = int(open(f"/sys/bus/pci/devices/{bdf}/numa_node").read())
        if node < 0:
            return None
        cpus = set()
        for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
            lo, _, hi = part.partition("-")
            cpus.update(range(int(lo), int(hi or lo) + 1))
        cpus &= os.sched_getaffinity(0)
        if not cpus:
            return None
        os.sched_setaffinity(0, cpus)
        return {"node": node, "cpus": len(cpus)}
    except (OSError, ValueError, AttributeError, RuntimeError):
        return None


def shard_range(n_streams: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous, balanced [begin, end) of rank's streams."""
    return (n_streams * rank) // world, (n_streams * (rank + 1)) // world


RECORD_I32 = 5   # nbytes, nbits, clock_idx, term_frame, status


def pack_records(res):
    """[n, stride + 20] uint8: decoded bytes row followed by the five int32 (little endian)."""
    import torch
    meta = torch.stack([res.nbytes, res.nbits, res.clock_idx, res.term_frame, res.status], dim=1)
    return torch.cat([res.bytes, meta.contiguous().view(torch.uint8).reshape(meta.shape[0], -1)], dim=1)


def unpack_records(rec, stride: int):
    import torch
    from .batch import DemodResult
    meta = rec[:, stride:].contiguous().view(torch.int32).reshape(rec.shape[0], RECORD_I32)
    return DemodResult(rec[:, :stride].contiguous(), meta[:, 0].contiguous(),
                       meta[:, 1].contiguous(), meta[:, 2].contiguous(), meta[:, 3].contiguous(),
                       meta[:, 4].contiguous())


def gather_flat(res, n_total: int, group=None, out=None):
    """Equal-shard fast path: ONE all_gather_into_tensor of ``res.flat`` (the single allocation
    behind a DemodResult from ``batch.alloc_result``) and nothing else -- no packing kernels.
    Returns a list of per-rank DemodResult views over the gathered buffer (rank r's streams are
    ``shard_range(n_total, r, world)``).  ``out``: optional preallocated uint8 [world * flat]."""
    import torch
    import torch.distributed as dist
    from .batch import views_of_flat
    world = dist.get_world_size(group)
    n_local, stride = int(res.bytes.shape[0]), int(res.bytes.shape[1])
    assert n_total == n_local * world, "gather_flat needs equal shards"
    flat = res.flat
    if out is None:
        out = torch.empty(world * flat.numel(), dtype=torch.uint8, device=flat.device)
    dist.all_gather_into_tensor(out, flat, group=group)
    return split_gathered(out, world, n_local, stride)


def gather_flat_to_root(res, n_total: int, dst: int = 0, group=None, out=None):
    """Equal shards, records wanted on ONE rank: every rank sends ``res.flat`` to rank ``dst``
    (``torch.distributed.gather``: over RCCL one direct transfer per rank into dst's xGMI links -- never a
    ring -- and 1/world of the all-gather's traffic).  ``dst`` is a rank OF THE GROUP (0 ... world - 1, the
    numbering ``shard_range`` uses; with the default group that is the global rank); it is translated to
    the global rank ``torch.distributed.gather`` expects.  Returns the list of per-rank DemodResult views on
    ``dst`` and None on the other ranks.  ``out``: optional preallocated uint8 [world * flat] on dst."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)                       # group-local, like dst
    if not 0 <= dst < world:
        raise ValueError(f"dst {dst} is not a rank of a group of {world}")
    dst_global = dst if group is None else dist.get_global_rank(group, dst)
    n_local, stride = int(res.bytes.shape[0]), int(res.bytes.shape[1])
    assert n_total == n_local * world, "gather_flat_to_root needs equal shards"
    flat = res.flat
    if rank != dst:
        dist.gather(flat, None, dst=dst_global, group=group)
        return None
    if out is None:
        out = torch.empty(world * flat.numel(), dtype=torch.uint8, device=flat.device)
    per = flat.numel()
    dist.gather(flat, [out[r * per: (r + 1) * per] for r in range(world)], dst=dst_global, group=group)
    return split_gathered(out, world, n_local, stride)


def split_gathered(out, world: int, n_local: int, stride: int):
    """Per-rank DemodResult views over a buffer filled by ``gather_flat`` (pure views; callers
    in a hot loop create them once for a preallocated ``out`` and reuse them)."""
    from .batch import views_of_flat
    per = out.numel() // world
    return [views_of_flat(out[r * per: (r + 1) * per], n_local, stride) for r in range(world)]


def gather_results(res, n_total: int, group=None):
    """All-gather every rank's DemodResult into one covering all n_total streams.

    Every rank must hold the shard ``shard_range(n_total, rank, world)``.  Shards
    may differ by one stream, so rows are padded to the largest shard for the
    single collective and trimmed afterwards.  (General form; ``gather_flat`` is the
    zero-copy form for equal shards.)
    """
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    stride = int(res.bytes.shape[1])
    rec = pack_records(res)
    b, e = shard_range(n_total, rank, world)
    assert rec.shape[0] == e - b, (rec.shape, b, e)
    max_rows = max(shard_range(n_total, r, world)[1] - shard_range(n_total, r, world)[0]
                   for r in range(world))
    if rec.shape[0] < max_rows:
        pad = torch.zeros((max_rows - rec.shape[0], rec.shape[1]), dtype=rec.dtype, device=rec.device)
        rec = torch.cat([rec, pad], dim=0)
    out = torch.empty((world * max_rows, rec.shape[1]), dtype=rec.dtype, device=rec.device)
    dist.all_gather_into_tensor(out, rec.contiguous(), group=group)
    parts = []
    for r in range(world):
        rb, re_ = shard_range(n_total, r, world)
        parts.append(out[r * max_rows: r * max_rows + (re_ - rb)])
    return unpack_records(torch.cat(parts, dim=0), stride)
