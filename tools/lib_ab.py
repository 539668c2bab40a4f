#!/usr/bin/env python3
"""A/B of several builds of libafsk_amd.so in ONE process, on one resident batch:

    python tools/lib_ab.py [--bauds 6000 | --bauds 375,160,96,1200] [--streams 65536] [--entry auto|mixed]
                           [--rounds 12] [--reps 5] [--order rotate] a.so b.so [c.so ...]

Every library is loaded with ctypes next to the others; the launches of one round go A, B, C, ... (with
--order rotate the starting library moves on every round, so that no build always runs behind the same
neighbour), each timed with HIP events on the launch stream; the outputs of every library are compared with the
first one's.  The batch is what bench.py builds for `--workload custom` (same payloads, same layout), synthesised
with the default library.  Passing the same file twice (a copy under another name) is the control: identical code
must read the same."""
from __future__ import annotations

import argparse
import ctypes as C
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="+")
    ap.add_argument("--bauds", default="1200")
    ap.add_argument("--streams", type=int, default=65536)
    ap.add_argument("--entry", default="auto", choices=["auto", "mixed"])
    ap.add_argument("--rounds", type=int, default=12)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--order", default="rotate", choices=["fixed", "rotate"])
    ap.add_argument("--rate-order", default="cycle", choices=["cycle", "blocks"], help="how --bauds are laid over the streams (bench.py --rate-order)")
    ap.add_argument("--offset-transpose", type=int, default=0,
                    help="K > 1: stream s reads the samples of stream (s %% K) * (n / K) + s / K -- a stream-order walk then jumps "
                         "through the input like a rate-sorted walk over K cycling rates (the round-trip figure is meaningless then)")
    ap.add_argument("--stride-align", type=int, default=0, help="round the output row stride up to a multiple of this (0: bench.py's)")
    ap.add_argument("--lead", default="", help="lead-in before every frame: N samples or 'random' (bench.py --lead)")
    a = ap.parse_args()
    import numpy as np
    import torch
    from afskmodem_amd import _native, batch
    bauds = tuple(int(b) for b in a.bauds.split(","))
    bench.WORKLOADS["custom"] = (a.streams, bauds, None, f"custom {bauds}")
    bench.RATE_ORDER = a.rate_order
    if a.lead == "random":
        bench.LEADS["custom"] = 2048
    elif a.lead:
        bench.LEAD_FIXED["custom"] = int(a.lead)
    args = types.SimpleNamespace(gpus=1, share_gpu0=False, force_gather=False, dist_backend="nccl", entry=a.entry,
                                 pg_timeout_s=90.0)
    os.environ.setdefault("AFSK_BENCH_VERBOSE", "0")
    ctx = bench.Ctx(args)
    sh = bench.Shard(ctx, "custom", a.streams)
    n, stride = sh.n_local, sh.stride
    if a.stride_align > 0:
        stride = -(-stride // a.stride_align) * a.stride_align
    if a.offset_transpose > 1:
        k = a.offset_transpose
        idx = torch.arange(n, device=ctx.dev)
        src = (idx % k) * (n // k) + idx // k
        sh.off = sh.off[src.clamp_(max=n - 1)].contiguous()
    sptr = C.c_void_p(ctx.cur.cuda_stream)
    libs, calls, outs = [], [], []
    for path in a.libs:
        path, _, win = path.partition("@")           # "lib.so@2048": AFSK_GROUP_WINDOW for this entry's group plan;
        win, _, sort_from = win.partition(":")       # "lib.so@4096:3": ... and AFSK_GROUP_SORT_FROM
        os.environ["AFSK_GROUP_WINDOW"] = win or "0"
        if sort_from:
            os.environ["AFSK_GROUP_SORT_FROM"] = sort_from
        else:
            os.environ.pop("AFSK_GROUP_SORT_FROM", None)
        L = C.CDLL(os.path.abspath(path))
        for name, (res, at) in _native.SIGNATURES.items():
            fn = getattr(L, name, None)                   # (an older build lacks the entries added since)
            if fn is not None:
                fn.restype, fn.argtypes = res, at
        o = batch.alloc_result(n, stride, ctx.dev)
        if sh.uniform_bf is not None:
            fn = L.afsk_demod_batch_uniform
            mk = lambda x, o=o: (x.data_ptr(), sh.off.data_ptr(), sh.ln.data_ptr(), sh.uniform_bf, 14000, n,   # noqa: E731
                                 o.bytes.data_ptr(), stride, o.nbytes.data_ptr(), o.nbits.data_ptr(), o.clock_idx.data_ptr(),
                                 o.term_frame.data_ptr(), o.status.data_ptr(), None, None, 0, sptr)
        elif a.entry != "mixed":
            h = C.c_void_p()
            assert L.afsk_group_plan_create(sh.bf_h.ctypes.data_as(C.POINTER(C.c_int32)), n, C.byref(h)) == 0
            fn = L.afsk_demod_batch_grouped
            mk = lambda x, o=o, h=h: (h, x.data_ptr(), sh.off.data_ptr(), sh.ln.data_ptr(), 14000, o.bytes.data_ptr(), stride,   # noqa: E731
                                      o.nbytes.data_ptr(), o.nbits.data_ptr(), o.clock_idx.data_ptr(), o.term_frame.data_ptr(),
                                      o.status.data_ptr(), None, None, 0, sptr)
        else:
            fn = L.afsk_demod_batch
            mk = lambda x, o=o: (x.data_ptr(), sh.off.data_ptr(), sh.ln.data_ptr(), sh.bf.data_ptr(), 14000, n,   # noqa: E731
                                 o.bytes.data_ptr(), stride, o.nbytes.data_ptr(), o.nbits.data_ptr(), o.clock_idx.data_ptr(),
                                 o.term_frame.data_ptr(), o.status.data_ptr(), sptr)
        libs.append(os.path.basename(path) + (("@" + win) if win else "") + ((":" + sort_from) if sort_from else ""))
        calls.append((fn, [mk(x) for x in sh.inputs]))
        outs.append(o)
    nin = len(sh.inputs)
    k = [0]

    def run(i: int, reps: int) -> float:
        fn, argl = calls[i]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(ctx.cur)
        for _ in range(reps):
            rc = fn(*argl[k[0] % nin])
            k[0] += 1
            assert rc == 0, rc
        e1.record(ctx.cur)
        e1.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3

    for i in range(len(libs)):                       # warm every library, compare outputs with the first one
        run(i, 3)
    torch.cuda.synchronize()
    ref = outs[0].cpu()
    same = []
    for o in outs:
        h = o.cpu()
        same.append(all(np.array_equal(getattr(h, f), getattr(ref, f)) for f in ("nbytes", "nbits", "clock_idx", "term_frame", "status"))
                    and h.payloads() == ref.payloads())
    for name, o, eq in zip(libs, outs, same):        # what differs, for the first few streams (debugging a variant)
        if eq:
            continue
        h = o.cpu()
        hp, rp = h.payloads(), ref.payloads()        # (once: a build whose every stream differs must not cost n^2)
        bad = [s_ for s_ in range(n) if any(getattr(h, f)[s_] != getattr(ref, f)[s_] for f in ("nbytes", "nbits", "clock_idx", "term_frame", "status"))
               or hp[s_] != rp[s_]]
        print(f"  {name}: {len(bad)} of {n} streams differ; first: " + "; ".join(
            f"s{s_}: " + ",".join(f"{f}={int(getattr(h, f)[s_])}/{int(getattr(ref, f)[s_])}" for f in ("nbytes", "nbits", "clock_idx", "term_frame", "status"))
            for s_ in bad[:4]))
    ok_payload = sum(p == sh.payload_h[s, : sh.plen_h[s]].tobytes() for s, p in enumerate(ref.payloads())) / n
    for _ in range(60):                              # pre-roll: settled clocks
        run(0, 1)
    times = [[] for _ in libs]
    for r in range(a.rounds):
        order = list(range(len(libs)))
        if a.order == "rotate":
            order = order[r % len(libs):] + order[: r % len(libs)]
        for i in order:
            t = run(i, a.reps)
            if r > 0:
                times[i].append(t)
    print(f"bauds {a.bauds} streams {n} entry {fn.__name__} out_stride {stride} round trip {ok_payload:.4f}")
    base = None
    for name, ts, eq in zip(libs, times, same):
        ts.sort()
        med = ts[len(ts) // 2]
        base = base or med
        print(f"  {name:28s} median {med:9.2f} us  min {ts[0]:9.2f} us  vs first {med / base:6.3f}  outputs {'==' if eq else '!='} first")


if __name__ == "__main__":
    main()
