"""world_size-2 gloo test of the shard + gather path (CPU tensors stand in for HBM)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from afskmodem_amd import dist as adist
from afskmodem_amd.batch import DemodResult


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _fake_result(begin, end, stride):
    n = end - begin
    idx = torch.arange(begin, end, dtype=torch.int32)
    b = (idx[:, None] * 3 + torch.arange(stride, dtype=torch.int32)[None, :]) % 251
    return DemodResult(b.to(torch.uint8), idx % 35, idx * 14, idx % 4096, idx + 24160, idx % 3)


def _flat_result(begin, end, stride):
    """Same content as _fake_result but living in one flat allocation (batch.alloc_result layout)."""
    from afskmodem_amd import batch
    src = _fake_result(begin, end, stride)
    res = batch.alloc_result(end - begin, stride, "cpu")
    for f in ("bytes", "nbytes", "nbits", "clock_idx", "term_frame", "status"):
        getattr(res, f).copy_(getattr(src, f))
    return res


def _worker_flat(rank, world, port, n_total, stride, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    b, e = adist.shard_range(n_total, rank, world)
    parts = adist.gather_flat(_flat_result(b, e, stride), n_total)
    ok = len(parts) == world
    for r, part in enumerate(parts):
        rb, re_ = adist.shard_range(n_total, r, world)
        want = _fake_result(rb, re_, stride)
        ok = ok and all(torch.equal(getattr(part, f), getattr(want, f))
                        for f in ("bytes", "nbytes", "nbits", "clock_idx", "term_frame", "status"))
    q.put((rank, ok, n_total))
    dist.destroy_process_group()


def _worker_root(rank, world, port, n_total, stride, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    b, e = adist.shard_range(n_total, rank, world)
    parts = adist.gather_flat_to_root(_flat_result(b, e, stride), n_total, dst=1)     # a root that is not rank 0
    if rank != 1:
        ok = parts is None
    else:
        ok = len(parts) == world
        for r, part in enumerate(parts):
            rb, re_ = adist.shard_range(n_total, r, world)
            want = _fake_result(rb, re_, stride)
            ok = ok and all(torch.equal(getattr(part, f), getattr(want, f))
                            for f in ("bytes", "nbytes", "nbits", "clock_idx", "term_frame", "status"))
    q.put((rank, ok, n_total))
    dist.destroy_process_group()


def _worker_subgroup(rank, world, port, n_total, stride, q):
    """a NON-default group whose local ranks differ from the global ones -- ranks {1, 2} of a world of 3.
    ``dst`` is a rank of the group (like shard_range's): dst = 1 is global rank 2."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    grp = dist.new_group([1, 2])                         # every rank of the world calls this
    ok = True
    if rank in (1, 2):
        g_rank, g_world = dist.get_rank(grp), dist.get_world_size(grp)
        b, e = adist.shard_range(n_total, g_rank, g_world)
        parts = adist.gather_flat_to_root(_flat_result(b, e, stride), n_total, dst=1, group=grp)
        if g_rank != 1:
            ok = parts is None
        else:
            ok = rank == 2 and len(parts) == g_world
            for r, part in enumerate(parts):
                rb, re_ = adist.shard_range(n_total, r, g_world)
                want = _fake_result(rb, re_, stride)
                ok = ok and all(torch.equal(getattr(part, f), getattr(want, f))
                                for f in ("bytes", "nbytes", "nbits", "clock_idx", "term_frame", "status"))
        full = adist.gather_flat(_flat_result(b, e, stride), n_total, group=grp)
        ok = ok and len(full) == g_world
        try:
            adist.gather_flat_to_root(_flat_result(b, e, stride), n_total, dst=2, group=grp)
            ok = False
        except ValueError:
            pass
    q.put((rank, ok, n_total))
    dist.barrier()
    dist.destroy_process_group()


def _worker(rank, world, port, n_total, stride, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    b, e = adist.shard_range(n_total, rank, world)
    full = adist.gather_results(_fake_result(b, e, stride), n_total)
    want = _fake_result(0, n_total, stride)
    ok = all(torch.equal(getattr(full, f), getattr(want, f))
             for f in ("bytes", "nbytes", "nbits", "clock_idx", "term_frame", "status"))
    q.put((rank, ok, int(full.bytes.shape[0])))
    dist.destroy_process_group()


def test_shard_range_partitions():
    for n in (0, 1, 7, 4096, 524288, 1001):
        for world in (1, 2, 3, 8):
            ranges = [adist.shard_range(n, r, world) for r in range(world)]
            assert ranges[0][0] == 0 and ranges[-1][1] == n
            for (b0, e0), (b1, e1) in zip(ranges, ranges[1:]):
                assert e0 == b1
            sizes = [e - b for b, e in ranges]
            assert max(sizes) - min(sizes) <= 1


def test_pack_unpack_roundtrip():
    r = _fake_result(5, 37, 40)
    rec = adist.pack_records(r)
    assert rec.shape == (32, 40 + 20)
    back = adist.unpack_records(rec, 40)
    for f in ("bytes", "nbytes", "nbits", "clock_idx", "term_frame", "status"):
        assert torch.equal(getattr(back, f), getattr(r, f))


def test_flat_layout_views():
    from afskmodem_amd import batch
    res = batch.alloc_result(7, 12, "cpu")
    offs, total = batch.flat_layout(7, 12)
    assert res.flat.numel() == total == 84 + 5 * 28
    res.nbits.fill_(3)
    res.bytes.fill_(9)
    assert int(res.flat[offs[2]: offs[2] + 4].view(torch.int32)[0]) == 3
    assert res.flat[:84].eq(9).all() and res.status.eq(0).all()


@pytest.mark.parametrize("n_total,worker", [(64, "pad"), (33, "pad"), (64, "flat"), (64, "root")])
def test_gather_world2_gloo(n_total, worker):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    target = {"pad": _worker, "flat": _worker_flat, "root": _worker_root}[worker]
    procs = [ctx.Process(target=target, args=(r, 2, port, n_total, 40, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == [0, 1]
    assert all(ok and rows == n_total for _, ok, rows in res)


def test_gather_to_root_in_a_subgroup_world3_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_subgroup, args=(r, 3, port, 64, 40, q)) for r in range(3)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == [0, 1, 2] and all(ok for _, ok, _ in res)
