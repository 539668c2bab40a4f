#!/usr/bin/env python3
"""bench.py -- batched AFSK demodulation throughput on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path (afsk_demod_batch: sync search + symbol correlator +
squelch + Hamming decode + byte pack) over one batch of synthetic streams already resident in
HBM.  Prints ONE JSON line (contract in the task statement).

Workloads (BASELINE.json configs[1..4]):
  config2  4096 x 1 s @1200 baud, clean            -- the headline workload at N = 1
  config3  65536 x 1 s mixed {300,1200,2400} baud  -- sub-record at N = 1
  config4  65536 x 1 s @1200 baud + noise          -- sub-record at N = 1: timed at 10 dB, plus
                                                      the BER curve 30 -> 5 dB vs the CPU oracle
  config5  65536 x 1 s @1200 baud per GPU          -- the headline workload at N > 1 (the shard
                                                      north_star names: 524288 streams on 8 GPUs);
                                                      sub-record at N = 1 (1-GPU point of its curve)
At N > 1 the config2 shard is carried as a sub-record, so both weak-scaling curves (4096 and
65536 streams per GPU) can be read off the N = 1, 2, 4, 8 lines (`per_workload_value`).

Launching: `python bench.py --gpus N` with N > 1 from a bare interpreter starts N rank processes
itself (torch.distributed.run on a free port) BEFORE anything touches the GPU and relays rank 0's
line and a non-zero exit code if any rank fails; under an external torchrun (RANK / WORLD_SIZE in
the environment) it runs as one rank.  N = 1 stays in-process (rocprofv3 ... -- python3 bench.py).

The line carries `roofline` (HBM-read bound; algorithmic bytes / HIP-event kernel time) and, at
N = 1, `cpu_baseline` (the CPU oracle -- a C port of the reference -- timed on this box's host
cores on a bounded sample, also the match-rate checker).  oracle/ is used as the checker only,
after the timed regions.
"""
from __future__ import annotations

import argparse
import ctypes as C
import hashlib
import json
import os
import random
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (streams per GPU, bauds cycled over streams, snr_db or None, description)
    "config2": (4096, (1200,), None, "configs[1]: 4096 streams x 1 s @1200 baud, clean, per GPU"),
    "config3": (65536, (300, 1200, 2400), None, "configs[2]: 65536 streams x 1 s mixed baud {300,1200,2400}, clean, per GPU"),
    "config4": (65536, (1200,), 10.0, "configs[3]: 65536 streams x 1 s @1200 baud, additive noise SNR 10 dB, per GPU"),
    "config5": (65536, (1200,), None, "configs[4]: 65536 streams x 1 s @1200 baud, clean, per GPU (524288 on 8)"),
    # not a BASELINE config: --workload custom --bauds 480,12000 [--streams N] times any baud mix
    # (profiles of the rates furthest from the roofline)
    "custom": (4096, (1200,), None, "custom: streams x 1 s, clean, bauds from --bauds, per GPU"),
}
BER_SNRS = (30, 25, 20, 15, 10, 7, 5, 3, 0)   # configs[3] sweep 30 -> 5 dB + two points below it
STREAM_LEN = 48000
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
MALL_BYTES = 256 << 20  # Infinity Cache: inputs smaller than a few of these are rotated
METRIC = "Msamples/s demodulated (batched 48 kHz streams) + decoded-byte match rate vs CPU ref"
PAYLOAD_SEED = 2024


def kernel_source_hash() -> str:
    """sha256 over the demod kernel's sources (afsk_demod*.h/.hip + afsk_kernels.h): ties a
    committed rocprof figure (profiles/traffic_latest.json) to the kernel it was measured on."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "afskmodem_amd", "csrc")
    for fn in sorted(os.listdir(d)):
        if fn.endswith((".h", ".hip")) and (fn.startswith("afsk_demod") or fn == "afsk_kernels.h"):
            h.update(fn.encode())
            h.update(open(os.path.join(d, fn), "rb").read())
    return h.hexdigest()[:16]


def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch(n: int, argv: list[str], script: str | None = None) -> int:
    """Parent of an N > 1 run: never touches the GPU, starts N fresh rank processes, relays
    rank 0's JSON line; non-zero exit if any rank failed or no line came back.
    (`script` is this file; tests/test_bench_launch.py passes a stub to exercise the relay.)"""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC for RCCL on this pool
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           script or os.path.abspath(__file__)] + argv
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    assert p.stdout is not None
    for ln in p.stdout:
        if ln.startswith('{"metric"'):
            line = ln.strip()
        else:
            sys.stderr.write(ln)
    rc = p.wait()
    if line is not None:
        print(line, flush=True)
    if rc == 0 and line is None:
        sys.stderr.write("bench.py: the ranks exited without a result line\n")
        rc = 1
    return rc


# --------------------------------------------------------------------------- one rank


class Ctx:
    """Per-process state: device, streams, process group."""

    def __init__(self, args):
        import torch
        import torch.distributed as dist
        from afskmodem_amd import _native
        self.torch, self.dist, self.args = torch, dist, args
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if self.world != args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={self.world}")
        _native.require_device()          # no GPU -> loud failure, never a CPU fallback
        # --share-gpu0 (diagnostic, with --dist-backend gloo): every rank uses device 0, so the whole
        # N > 1 flow (launcher, ranks, gather, per-rank checks) can be exercised on a 1-GPU box;
        # RCCL itself refuses two ranks on one device
        dev_index = 0 if args.share_gpu0 else self.local_rank
        torch.cuda.set_device(dev_index)
        self.dev = torch.device("cuda", dev_index)
        self.use_dist = self.world > 1 or args.force_gather
        if self.use_dist:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if "MASTER_PORT" not in os.environ:
                os.environ["MASTER_PORT"] = str(free_port())   # single-rank --force-gather only
            if args.dist_backend == "nccl":
                dist.init_process_group("nccl", rank=self.rank, world_size=self.world, device_id=self.dev)
            else:
                dist.init_process_group(args.dist_backend, rank=self.rank, world_size=self.world)
        self.backend = args.dist_backend
        self.lib = _native.lib()
        self.cur = torch.cuda.current_stream()
        self.comm = torch.cuda.Stream(device=self.dev) if self.use_dist else None

    def t(self, a):
        return self.torch.from_numpy(np.ascontiguousarray(a)).to(self.dev)

    def all_gather(self, out, inp) -> None:
        """all_gather_into_tensor on device tensors (RCCL); the gloo diagnostic backend stages
        through host memory."""
        if self.backend == "nccl":
            self.dist.all_gather_into_tensor(out, inp)
            return
        h_out = self.torch.empty(out.shape, dtype=out.dtype)
        self.dist.all_gather_into_tensor(h_out, inp.cpu())
        out.copy_(h_out)

    def all_reduce(self, t, op=None) -> None:
        kw = {} if op is None else {"op": op}
        if self.backend == "nccl":
            self.dist.all_reduce(t, **kw)
            return
        h = t.cpu()
        self.dist.all_reduce(h, **kw)
        t.copy_(h)

    def fence(self) -> None:
        self.torch.cuda.synchronize()
        if self.use_dist:
            self.dist.barrier()
            self.torch.cuda.synchronize()


class Shard:
    """This rank's contiguous range of one workload's streams, synthesised on the device."""

    def __init__(self, ctx: Ctx, name: str, n_local: int, snr_db=None, copies: int = 0):
        from afskmodem_amd import batch, synth
        from afskmodem_amd import dist as adist
        torch = ctx.torch
        _, bauds, wl_snr, desc = WORKLOADS[name]
        self.ctx, self.name, self.desc, self.bauds = ctx, name, desc, bauds
        self.snr_db = wl_snr if snr_db is None else snr_db
        self.n_local = n_local
        self.n_total = n_local * ctx.world
        self.first = ctx.rank * n_local
        assert adist.shard_range(self.n_total, ctx.rank, ctx.world) == (self.first, self.first + n_local)
        self.bf_h, self.plen_h, self.payload_h, ts_h = self.host_meta(self.first, n_local, bauds)
        self.off, self.ln = batch.uniform_layout(n_local, STREAM_LEN, ctx.dev)
        self.bf = ctx.t(self.bf_h)
        self._payload_d, self._plen_d, self._ts_d = ctx.t(self.payload_h), ctx.t(self.plen_h), ctx.t(ts_h)
        x = torch.empty(n_local * STREAM_LEN, dtype=torch.int16, device=ctx.dev)
        self.inputs = [x]
        self.regenerate(self.snr_db)
        # Inputs that could (partly) survive in the 256 MiB Infinity Cache between launches are
        # rotated over several distinct copies, so no step re-reads what the previous one fetched.
        nbytes = x.numel() * 2
        if copies <= 0:
            copies = 1 if nbytes >= 8 * MALL_BYTES else max(3, -(-6 * MALL_BYTES // nbytes))
            copies = min(copies, 8)
        for _ in range(copies - 1):
            self.inputs.append(x.clone())
        self.stride = batch.out_stride_for(STREAM_LEN, int(self.bf_h.min()))
        _, self.flat_sz = batch.flat_layout(n_local, self.stride)
        torch.cuda.synchronize()

    @staticmethod
    def host_meta(first: int, n: int, bauds):
        from afskmodem_amd import synth
        gidx = np.arange(first, first + n)
        baud_arr = np.asarray([bauds[i % len(bauds)] for i in gidx], np.int32)
        bf_h = (48000 // baud_arr).astype(np.int32)
        plen_h = np.asarray([synth.one_second_payload(int(b)) for b in baud_arr], np.int32)
        pstride = max(synth.one_second_payload(int(b)) for b in bauds)
        payload_h = synth.payload_bytes(PAYLOAD_SEED, first, n, pstride)
        ts_h = np.asarray([synth.ts_cycles_for(int(b)) for b in baud_arr], np.int32)
        return bf_h, plen_h, payload_h, ts_h

    def regenerate(self, snr_db, seed: int = 99) -> None:
        """(Re)write inputs[0]: Transmitter frames (+ additive noise at snr_db)."""
        from afskmodem_amd import batch, synth
        x = self.inputs[0]
        # the .wav writer's decimate/duplicate quirk (ref:239-244) is part of every Transmitter.save
        # stream; at 12000 baud it destroys the mark tone (in the reference too), so a custom
        # workload with that rate uses the ideal frames
        batch.modulate_batch(self._payload_d, self._plen_d, self.bf, self._ts_d, self.off, self.ln,
                             STREAM_LEN, x, 12000 not in self.bauds)
        if snr_db is not None:
            batch.add_noise_batch(x, self.off, self.ln, STREAM_LEN, synth.snr_to_scale_q24(snr_db),
                                  seed=seed, stream_idx_base=self.first)
        self.ctx.torch.cuda.synchronize()


def weighted_sum(torch, flat):
    """Order-sensitive checksum of a uint8 tensor (int64 arithmetic on the device)."""
    w = (torch.arange(flat.numel(), device=flat.device, dtype=torch.int64) % 65521) + 1
    return (flat.to(torch.int64) * w).sum()


def measure(ctx: Ctx, sh: Shard, steps: int, warmup: int, preroll_ms: float, gather_every: int = 0):
    """Pre-roll, W warm-up steps, then exactly K timed steps between two fences.  Every step
    writes its own output slot; with a process group the slots of G consecutive steps are
    all-gathered by ONE collective on a side stream while the next group runs."""
    from afskmodem_amd import _native, batch
    torch, dist = ctx.torch, ctx.dist
    cur, comm = ctx.cur, ctx.comm
    n_local, stride, flat_sz = sh.n_local, sh.stride, sh.flat_sz
    world, rank = ctx.world, ctx.rank

    # Each collective costs the compute stream ~40 us (cross-stream events around it),
    # so a group should cover a few ms of kernels: 64 steps of config #2, 3-4 of the 6.29 GB ones.
    if gather_every > 0:
        G = gather_every
    else:
        est_step_s = 2.0 * n_local * STREAM_LEN / 6.0e12
        G = max(1, min(64, int(4e-3 / est_step_s)))
    if comm is None:
        G = 1
    # one output slot per step (so any step can be checked afterwards), at most 256, whole groups
    nslots = max(2 * G, min(256, -(-max(steps, 1) // G) * G))
    nslots = -(-nslots // G) * G
    ngroups = nslots // G
    out_all = torch.zeros(nslots * flat_sz, dtype=torch.uint8, device=ctx.dev)
    slots = [batch.views_of_flat(out_all[k * flat_sz: (k + 1) * flat_sz], n_local, stride) for k in range(nslots)]
    gath_bufs = ([torch.zeros(world * G * flat_sz, dtype=torch.uint8, device=ctx.dev) for _ in range(2)]
                 if comm is not None else None)
    ready_ev = [torch.cuda.Event() for _ in range(ngroups)]
    done_ev = [torch.cuda.Event() for _ in range(ngroups)]
    gathered_once = [False] * ngroups

    sptr = C.c_void_p(cur.cuda_stream)
    fn = ctx.lib.afsk_demod_batch
    nin = len(sh.inputs)
    # argument tuples are built once: the timed loop is one ctypes call per step
    slot_args = [[(x.data_ptr(), sh.off.data_ptr(), sh.ln.data_ptr(), sh.bf.data_ptr(), 14000, n_local,
                   o.bytes.data_ptr(), stride, o.nbytes.data_ptr(), o.nbits.data_ptr(),
                   o.clock_idx.data_ptr(), o.term_frame.data_ptr(), o.status.data_ptr(), sptr)
                  for o in slots] for x in sh.inputs]

    def launch(i: int) -> None:
        rc = fn(*slot_args[i % nin][i % nslots])
        if rc != 0:
            _native.check(rc)

    gather_timing: list = []
    state = {"timing": False, "gathers": 0, "last_group": -1}

    def gather_group(g: int) -> None:
        gr = g % ngroups
        ready_ev[gr].record(cur)
        comm.wait_event(ready_ev[gr])
        with torch.cuda.stream(comm):
            pair = None
            if state["timing"] and len(gather_timing) < 256:
                pair = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                pair[0].record(comm)
            ctx.all_gather(gath_bufs[g & 1], out_all[gr * G * flat_sz: (gr + 1) * G * flat_sz])
            if pair is not None:
                pair[1].record(comm)
                gather_timing.append(pair)
            done_ev[gr].record(comm)
        gathered_once[gr] = True
        state["gathers"] += 1
        state["last_group"] = g

    def step(i: int) -> None:
        """Launch step i into slot i % nslots; after the last slot of a group, gather the group
        on the comm stream (it overlaps the next group's kernels)."""
        if comm is not None and i % G == 0:
            gr = (i // G) % ngroups
            if gathered_once[gr]:
                cur.wait_event(done_ev[gr])     # the collective that read these slots last time is done
        launch(i)
        if comm is not None and i % G == G - 1:
            gather_group(i // G)

    def finish(n_steps: int) -> None:
        """Gather a trailing partial group so that every step's records have been exchanged."""
        if comm is not None and n_steps % G != 0:
            gather_group((n_steps - 1) // G)

    # Pre-roll: the device needs ~20 ms of sustained load before its clocks settle (profiles/
    # README.md).  Same kernel, same buffers, not timed; then the W warm-up steps.
    preroll_launches = 0
    if preroll_ms > 0:
        t_pre = time.perf_counter()
        for i in range(4):
            launch(i)
        torch.cuda.synchronize()
        est = max((time.perf_counter() - t_pre) / 4, 1e-5)
        preroll_launches = int(preroll_ms * 1e-3 / est) + 1
        for i in range(preroll_launches):
            launch(i)
        torch.cuda.synchronize()
    for i in range(warmup):
        step(i)
    finish(warmup)
    torch.cuda.synchronize()
    if comm is not None:
        comm.synchronize()
    out_all.zero_()            # a correct slot after the timed region was written BY the timed region
    for gr in range(ngroups):
        gathered_once[gr] = False
    ctx.fence()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    state["timing"] = True
    t0 = time.perf_counter()
    ev0.record(cur)
    for i in range(steps):
        step(i)
    finish(steps)
    ev1.record(cur)
    host_issue_s = time.perf_counter() - t0        # host time to enqueue the whole timed region
    ctx.fence()
    elapsed = time.perf_counter() - t0
    event_ms = ev0.elapsed_time(ev1)
    kernel_ms = event_ms / max(steps, 1)           # avg launch duration incl. any gaps
    if ctx.use_dist:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=ctx.dev)
        ctx.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # ---- which step gets the full check: a random one among those whose slot still holds it
    lo = max(0, steps - nslots)
    chk_step = random.SystemRandom().randrange(lo, steps) if steps > 0 else 0
    chk = slots[chk_step % nslots]
    used = min(steps, nslots)
    rows = out_all[: used * flat_sz].view(used, flat_sz)
    all_steps_equal = bool((rows == rows[chk_step % nslots]).all().item()) if used > 0 else None
    res = chk.cpu()
    got_payloads = res.payloads()
    if sh.snr_db is None:
        ok = sum(got_payloads[s] == sh.payload_h[s, : sh.plen_h[s]].tobytes() for s in range(n_local))
        roundtrip_rate = ok / n_local
    else:
        roundtrip_rate = None
    # samples the reference must read: up to and including the squelch-triggering symbol
    active = np.minimum(np.maximum(res.term_frame.astype(np.int64)
                                   + (res.nbits.astype(np.int64) + 1) * sh.bf_h, 4096), STREAM_LEN)
    out_bytes_alg = int(np.minimum(res.nbytes, stride).sum()) + 20 * n_local
    alg_bytes = int(2 * active.sum()) + out_bytes_alg
    achieved_gbs = alg_bytes / (kernel_ms * 1e-3) / 1e9
    samples_per_step = sh.n_total * STREAM_LEN
    rec = {
        "workload": sh.desc,
        "streams_per_gpu": n_local,
        "steps": steps, "warmup": warmup, "preroll_launches": preroll_launches,
        "value": round(samples_per_step * steps / elapsed / 1e6, 1),            # wall clock, fences included
        "value_event_time": round(samples_per_step * steps / (event_ms * 1e-3) / 1e6, 1),   # HIP events
        # the same wall-clock rate counting only the samples the reference has to read (up to the
        # squelch-triggering symbol); `value` counts every sample of the buffers, tail silence included
        "value_active_samples": round(float(active.sum()) * ctx.world * steps / elapsed / 1e6, 1),
        "unit": "Msamples/s",
        "ms_per_step": round(elapsed / max(steps, 1) * 1e3, 5),
        "roofline": {"bound": "hbm", "achieved": round(achieved_gbs, 1), "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": round(achieved_gbs / HBM_PEAK_GBS, 4),
                     "traffic": None,
                     "algorithmic_bytes_per_launch": alg_bytes,
                     "kernel_ms": round(kernel_ms, 5),
                     "full_buffer_gbs": round((2 * n_local * STREAM_LEN) / (kernel_ms * 1e-3) / 1e9, 1)},
        "input_buffers_rotated": nin,
        "roundtrip_match_rate": roundtrip_rate,
        "checked_step": chk_step,
        "all_timed_steps_identical": all_steps_equal,
        "host_issue_ms_per_step": round(host_issue_s / max(steps, 1) * 1e3, 5),
    }
    aux = {"res": res, "got_payloads": got_payloads}

    if comm is not None:
        # Every rank's slice of the last gathered group, on every rank: (a) its checksum equals
        # the one its producer computed of its own buffer, (b) it decodes to that rank's payloads.
        comm.synchronize()
        g = state["last_group"]
        gr = g % ngroups
        gbuf = gath_bufs[g & 1]
        own = weighted_sum(torch, out_all[gr * G * flat_sz: (gr + 1) * G * flat_sz]).reshape(1)
        sums = torch.zeros(world, dtype=torch.int64, device=ctx.dev)
        ctx.all_gather(sums, own)
        k_last = (steps - 1) % G
        per_rank = []
        for r in range(world):
            part = gbuf[r * G * flat_sz: (r + 1) * G * flat_sz]
            ok_sum = bool((weighted_sum(torch, part) == sums[r]).item())
            v = batch.views_of_flat(part[k_last * flat_sz: (k_last + 1) * flat_sz], n_local, stride)
            ok_pay = None
            if sh.snr_db is None:
                _, plen_r, payload_r, _ = Shard.host_meta(r * n_local, n_local, sh.bauds)
                pl = ctx.t(plen_r)
                exp = ctx.t(payload_r)
                width = min(int(exp.shape[1]), stride)
                col = torch.arange(width, device=ctx.dev)[None, :]
                m = col < pl[:, None]
                same = ((v.bytes[:, :width] == exp[:, :width]) | ~m).all(dim=1) & (v.nbytes == pl)
                ok_pay = bool(same.all().item())
            per_rank.append(ok_sum and ok_pay is not False)
        mine = torch.tensor([int(all(per_rank))], dtype=torch.int32, device=ctx.dev)
        ctx.all_reduce(mine, op=dist.ReduceOp.MIN)
        ones = torch.ones(1, dtype=torch.int32, device=ctx.dev)
        ctx.all_reduce(ones, op=dist.ReduceOp.SUM)
        rec["ranks_seen"] = int(ones.item())
        rec["gather_check"] = per_rank                       # rank 0's view: one entry per producer rank
        rec["gather_check_on_every_rank"] = bool(mine.item())
        rec["gather_every_steps"] = G
        rec["gathers_in_timed_region"] = -(-steps // G)
        if gather_timing:
            # duration of the RCCL all-gather itself (comm stream, overlapped with the next
            # group's kernels), reported separately as SURVEY 8(d) config 5 asks
            gms = sorted(a.elapsed_time(b) for a, b in gather_timing)
            rec["gather_ms"] = {"median": round(gms[len(gms) // 2], 4), "max": round(gms[-1], 4),
                                "bytes_per_rank": int(G * flat_sz), "measured": len(gms)}
    del out_all, slots, gath_bufs
    return rec, aux


def oracle_match(sh: Shard, res, got_payloads, x_dev, ns: int, n_threads: int):
    """Decode the first ns streams of x_dev with the CPU oracle and compare every output."""
    from oracle import afsk_oracle as O   # checker only
    h = x_dev[: ns * STREAM_LEN].cpu().numpy()
    h_off = np.arange(ns, dtype=np.int64) * STREAM_LEN
    h_ln = np.full(ns, STREAM_LEN, np.int32)
    want = O.demod_batch(h, h_off, h_ln, sh.bf_h[:ns], 14000, out_stride=sh.stride, n_threads=n_threads)
    match = 0
    for s in range(ns):
        nb = int(want["nbytes"][s])
        m = min(nb, sh.stride)
        match += bool(nb == int(res.nbytes[s]) and int(want["nbits"][s]) == int(res.nbits[s])
                      and int(want["clock_idx"][s]) == int(res.clock_idx[s])
                      and int(want["term_frame"][s]) == int(res.term_frame[s])
                      and want["bytes"][s, :m].tobytes() == got_payloads[s][:m])
    return match / ns, want, (h, h_off, h_ln)


def ber_of(nbytes, out_bytes, payload, plen: int):
    """payload bit errors + 8 per missing/extra byte, over the given streams."""
    nb = nbytes.astype(np.int64)
    m = np.minimum(nb, plen)
    col = np.arange(plen)[None, :]
    diff = np.unpackbits((out_bytes[:, :plen] ^ payload[:, :plen]) * (col < m[:, None]).astype(np.uint8),
                         axis=1).sum(axis=1)
    errs = diff + 8 * np.abs(nb - plen)
    return float(errs.sum()) / (len(nb) * plen * 8), int((errs > 0).sum())


def ber_curve(ctx: Ctx, sh: Shard, sample: int, n_threads: int) -> list:
    """configs[3]: all streams of the shard at every SNR on the GPU; the CPU oracle decodes the
    first `sample` of them and must agree stream by stream, so the two curves coincide."""
    from afskmodem_amd import batch
    from oracle import afsk_oracle as O   # checker only
    torch = ctx.torch
    rows = []
    plen = int(sh.plen_h[0])
    ns = min(sample, sh.n_local)
    for snr in BER_SNRS:
        sh.regenerate(float(snr), seed=1000 + snr)
        res = batch.demod_batch(sh.inputs[0], sh.off, sh.ln, sh.bf, 14000, out_stride=sh.stride)
        torch.cuda.synchronize()
        got = res.cpu()
        ber, bad = ber_of(got.nbytes, got.bytes, sh.payload_h, plen)
        rate, want, _ = oracle_match(sh, got, got.payloads(), sh.inputs[0], ns, n_threads)
        ber_gpu_s, _ = ber_of(got.nbytes[:ns], got.bytes[:ns], sh.payload_h[:ns], plen)
        ber_cpu_s, _ = ber_of(want["nbytes"], want["bytes"], sh.payload_h[:ns], plen)
        rows.append({"snr_db": snr, "ber_gpu_all_streams": ber, "streams_with_errors": bad,
                     "over_read_streams": int((got.nbits > 14 * plen).sum()),
                     "ber_gpu_on_sample": ber_gpu_s, "ber_cpu_on_sample": ber_cpu_s,
                     "ber_equal": ber_gpu_s == ber_cpu_s, "cpu_match_rate": rate, "cpu_sample_streams": ns})
        del res
    return rows


def attach_traffic(rec: dict, name: str, n_local: int, src_hash: str) -> None:
    """roofline.traffic = HBM bytes per launch from the committed rocprofv3 PMC passes
    (profiles/traffic_latest.json) -- only when that profile was taken on THIS kernel source;
    never measured inside this run (PMC counters need the profiler), and labelled so."""
    rf = rec["roofline"]
    rf["traffic_source"] = None
    tfile = os.path.join(ROOT, "profiles", "traffic_latest.json")
    try:
        tj = json.load(open(tfile))
    except Exception:  # noqa: BLE001
        return
    ent = (tj.get("entries") or {}).get(name)
    if not ent or ent.get("streams") != n_local:
        return
    if tj.get("kernel_source_hash") != src_hash:
        rf["traffic_source"] = (f"none: profiles/traffic_latest.json was measured on kernel source "
                                f"{tj.get('kernel_source_hash')}, this build is {src_hash}")
        return
    rf["traffic"] = ent.get("hbm_bytes_per_launch")
    rf["traffic_source"] = (f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this kernel source "
                            f"({src_hash}) on another run: {ent.get('source')}; not measured in this run")


def run_rank(args) -> None:
    ctx = Ctx(args)
    torch, dist = ctx.torch, ctx.dist
    world, rank = ctx.world, ctx.rank
    cores = os.cpu_count() or 1
    src_hash = kernel_source_hash()

    main_name = args.workload or ("config2" if world == 1 else "config5")
    if args.sub is not None:
        sub_names = [s for s in args.sub.split(",") if s]
    elif args.workload or args.streams:
        sub_names = []
    else:
        sub_names = ["config3", "config4", "config5"] if world == 1 else ["config2"]
    for s in [main_name] + sub_names:
        if s not in WORKLOADS:
            raise SystemExit(f"unknown workload {s}")

    def n_for(name: str) -> int:
        return args.streams if args.streams > 0 else WORKLOADS[name][0]

    # ---------------- main record
    sh = Shard(ctx, main_name, n_for(main_name))
    rec, aux = measure(ctx, sh, args.steps, args.warmup, args.preroll_ms, args.gather_every)
    attach_traffic(rec, main_name, sh.n_local, src_hash)
    out = {
        "metric": METRIC,
        "value": rec["value"],
        "unit": "Msamples/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "preroll_launches": rec["preroll_launches"],
        "ms_per_step": rec["ms_per_step"],
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "int16",
        "data": "synthetic",
        "config": {"workload": sh.desc, "streams_per_gpu": sh.n_local, "streams_total": sh.n_total,
                   "stream_len": STREAM_LEN, "bauds": list(sh.bauds), "snr_db": sh.snr_db,
                   "parallelism": f"stream-sharded x{world}" + (" + RCCL all-gather of decoded records" if world > 1 else "")
                                  + (" [DIAGNOSTIC: gloo backend" + (", all ranks on one GPU" if args.share_gpu0 else "") + "]"
                                     if ctx.backend != "nccl" else "")},
        "value_event_time": rec["value_event_time"],
        "value_active_samples": rec["value_active_samples"],
        "roofline": rec["roofline"],
        "kernel_source_hash": src_hash,
        "input_buffers_rotated": rec["input_buffers_rotated"],
        "roundtrip_match_rate": rec["roundtrip_match_rate"],
        "checked_step": rec["checked_step"],
        "all_timed_steps_identical": rec["all_timed_steps_identical"],
        "host_issue_ms_per_step": rec["host_issue_ms_per_step"],
    }
    for k in ("ranks_seen", "gather_check", "gather_check_on_every_rank", "gather_every_steps",
              "gathers_in_timed_region", "gather_ms"):
        if k in rec:
            out[k] = rec[k]

    if world == 1 and not args.no_cpu_baseline:
        from oracle import afsk_oracle as O   # checker + reported CPU baseline only
        res, got_payloads = aux["res"], aux["got_payloads"]
        ns = args.cpu_sample_streams or min(sh.n_local, 4096)
        rate, _, (h, h_off, h_ln) = oracle_match(sh, res, got_payloads, sh.inputs[0], ns, cores)
        n1 = max(ns // 8, 1)
        t1 = time.perf_counter()
        O.demod_batch(h[: n1 * STREAM_LEN], h_off[:n1], h_ln[:n1], sh.bf_h[:n1], 14000,
                      out_stride=sh.stride, n_threads=1)
        dt1 = time.perf_counter() - t1
        reps = 0
        t2 = time.perf_counter()
        while True:
            O.demod_batch(h, h_off, h_ln, sh.bf_h[:ns], 14000, out_stride=sh.stride, n_threads=cores)
            reps += 1
            if time.perf_counter() - t2 > 10.0 or reps >= 50:
                break
        dtc = (time.perf_counter() - t2) / reps
        # "reference-shaped" figure: the pure-Python restatement (oracle/pyref.py) on one core;
        # in the dev container it runs at 1.01-1.07x the speed of the real afskmodem.py.
        from oracle import pyref
        npy = 3
        t3 = time.perf_counter()
        for s_i in range(npy):
            data, _, _, _ = pyref.demod(h[s_i * STREAM_LEN: (s_i + 1) * STREAM_LEN].tolist(), int(sh.bf_h[s_i]))
            assert data == got_payloads[s_i], "pure-Python restatement disagrees with the GPU"
        dtp = (time.perf_counter() - t3) / npy
        out["cpu_baseline"] = {
            "value": round(ns * STREAM_LEN / dtc / 1e6, 1), "unit": "Msamples/s", "cores": cores,
            "kind": "port",
            "sample": f"first {ns} streams of the same batch, CPU oracle (C port of afskmodem.py hot path), "
                      f"{cores} threads, {reps} reps; single thread on {n1} streams: "
                      f"{round(n1 * STREAM_LEN / dt1 / 1e6, 1)} Msamples/s",
            "single_thread_value": round(n1 * STREAM_LEN / dt1 / 1e6, 1),
            "python_reference_shaped_value": round(STREAM_LEN / dtp / 1e6, 3),
            "python_reference_shaped_note": "oracle/pyref.py (pure-Python restatement, CPython, 1 core, "
                                            f"{npy} streams); calibrated at 1.01-1.07x the real reference's "
                                            "speed in the dev container (DESIGN.md 4.3); the real afskmodem.py "
                                            "(1.3 Msamples/s/core) was only ever timed in the build container",
        }
        out["match_rate"] = rate
        out["match_sample_streams"] = ns
        del h
    del sh, aux
    torch.cuda.empty_cache()

    # ---------------- sub-records: the other single-GPU configs in the same line
    per_workload = {main_name: out["value"]}
    subs = {}
    for name in sub_names:
        if name == main_name:
            continue
        n_local = n_for(name)
        big = n_local * STREAM_LEN * 2 >= (1 << 30)
        s_steps = args.sub_steps or (20 if big else 200)
        s_warm = 3 if big else 20
        shs = Shard(ctx, name, n_local)
        srec, saux = measure(ctx, shs, s_steps, s_warm, min(args.preroll_ms, 100.0), args.gather_every)
        attach_traffic(srec, name, n_local, src_hash)
        if world == 1 and not args.no_cpu_baseline:
            ns = min(n_local, args.sub_cpu_sample)
            srec["match_rate"], _, _ = oracle_match(shs, saux["res"], saux["got_payloads"], shs.inputs[0], ns, cores)
            srec["match_sample_streams"] = ns
            if name == "config4":
                srec["ber_curve"] = ber_curve(ctx, shs, ns, cores)
        per_workload[name] = srec["value"]
        subs[name] = srec
        del shs, saux
        torch.cuda.empty_cache()
    if subs:
        out["sub_records"] = subs
    out["per_workload_value"] = per_workload
    out["scaling_note"] = ("headline workload: config2 (4096 streams/GPU) at N = 1, config5 (65536 streams/GPU, the shard "
                           "north_star names) at N > 1; every line carries both in per_workload_value, so each "
                           "weak-scaling curve has its own 1-GPU point (config5's is per_workload_value.config5 of the N = 1 line)")
    if ctx.use_dist:
        dist.destroy_process_group()
    # RCCL writes its version banner through C stdio, which would otherwise be flushed at exit,
    # AFTER the result: flush it now so that the JSON line is the last thing on stdout
    try:
        C.CDLL(None).fflush(None)
    except Exception:  # noqa: BLE001
        pass
    if rank == 0:
        print(json.dumps(out), flush=True)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="", choices=[""] + sorted(WORKLOADS),
                    help="headline workload (default: config2 at N = 1, config5 at N > 1); "
                         "giving one explicitly drops the sub-records unless --sub lists them")
    ap.add_argument("--sub", default=None, help="comma list of sub-record workloads ('' = none)")
    ap.add_argument("--sub-steps", type=int, default=0, help="timed steps of every sub-record (0 = 20 / 200)")
    ap.add_argument("--sub-cpu-sample", type=int, default=1024,
                    help="streams per sub-record (and per SNR of the BER curve) decoded by the CPU oracle")
    ap.add_argument("--streams", type=int, default=0, help="override streams per GPU")
    ap.add_argument("--bauds", default="", help="with --workload custom: comma list of baud rates cycled over the streams")
    ap.add_argument("--preroll-ms", type=float, default=300.0,
                    help="untimed launches of the same kernel before the warm-up steps, so the "
                         "GPU clocks have settled (the first ~20 ms under load run 4-6 %% slower)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-streams", type=int, default=0)
    ap.add_argument("--gather-every", type=int, default=0,
                    help="N > 1: all-gather the decoded records of G steps with one collective; "
                         "0 = as many steps as take about 4 ms (at most 64)")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (the product); gloo = diagnostic, collectives staged through the host")
    ap.add_argument("--share-gpu0", action="store_true",
                    help="diagnostic: all ranks on device 0 (needs --dist-backend gloo), to exercise the N > 1 flow on a 1-GPU box")
    ap.add_argument("--force-gather", action="store_true",
                    help="exercise the RCCL gather path even at N=1 (single-rank group); diagnostics")
    args = ap.parse_args()

    if args.bauds:
        bl = tuple(int(b) for b in args.bauds.split(","))
        WORKLOADS["custom"] = (WORKLOADS["custom"][0], bl, None, f"custom: streams x 1 s, clean, bauds {list(bl)}, per GPU")
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        # bare `python bench.py --gpus N`: become the launcher.  Nothing above or below this line
        # touches the GPU in this process (device_count() does not initialise it on this image).
        import torch
        have = torch.cuda.device_count()
        if args.share_gpu0 and args.dist_backend != "gloo":
            sys.stderr.write("bench.py: --share-gpu0 needs --dist-backend gloo (RCCL refuses two ranks on one device)\n")
            raise SystemExit(2)
        if have < (1 if args.share_gpu0 else args.gpus):
            sys.stderr.write(f"bench.py: --gpus {args.gpus} but only {have} GPU(s) visible\n")
            raise SystemExit(2)
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))
    run_rank(args)


if __name__ == "__main__":
    main()
