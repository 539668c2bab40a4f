#!/usr/bin/env python3
"""bench.py -- batched AFSK demodulation throughput on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path (sync search + symbol correlator + squelch + Hamming decode +
byte pack: afsk_demod_batch_uniform for a one-rate batch, afsk_demod_batch for a mixed one) over one
batch of synthetic streams already resident in HBM.  Prints ONE JSON line (contract in the task
statement).

The HEADLINE workload is the same at every N: the config5 shard -- BASELINE.json configs[4],
65536 streams x 1 s @1200 baud PER GPU (524288 on 8), the largest single-GPU configuration and the
one north_star names -- so the N = 1, 2, 4, 8 lines form one weak-scaling curve and value(1) of a
scaling run equals the N = 1 bench line.  At N = 1 the other configs ride along as `sub_records`:
  config2  4096 x 1 s @1200 baud, clean (configs[1])
  config3  65536 x 1 s mixed {300,1200,2400} baud (configs[2]), with a CPU baseline per baud share
  config4  65536 x 1 s @1200 baud + noise (configs[3]): timed at 10 dB, plus the BER curve 30 -> 0 dB
           vs the CPU oracle
and so do the three built "next" rows of SURVEY 8(f), each with its own roofline and oracle check:
  f1_modulate    on-device modulator (write-bound, config5 shape)
  f2_gate        live-gate replay (read-bound) over 4096 and 65536 captures
  f3_wav_ingest  4096 .wav files -> device layout (PCIe-bound: against a pinned hipMemcpy of the bytes)
  f5_wav_egress  4096 device streams -> .wav files (the mirror; against a pinned device-to-host hipMemcpy)
plus `rates_4096` / `rates_65536`: 4096 and 65536 x 1 s at each of the 36 rates a Receiver can be built for
(12000 ... 24 baud; the second is the steady-state fraction per rate, free of launch-shape quantisation).
At N > 1 the config2 shard is carried as a sub-record (worst case for the per-collective cost).

Output: ONE line on stdout, under 4 KB (`compact_line`: the contract's keys, `roofline`, `cpu_baseline`, match
rates, {value, frac, match_rate} per sub-record); the full record (per-rate tables, BER rows, event intervals:
tens of KB) goes to gpurun_out/bench_full_n<N>.json, named in the line as `full_record`.  Nothing large goes to
stderr either: the driver keeps one ~8 KB tail of both streams.

Timing: a timed region is EXACTLY K steps between two fences (barrier + synchronize on both sides);
when one region is shorter than --min-region-ms (50 ms) the K-step region is repeated and `value` /
`ms_per_step` are those of the MEDIAN region (`timed_regions`, `timed_region_ms`).  HIP events inside the
regions give the per-step median (`event_ms_per_step_median`) and the average launch duration of the
SAME median region, which the roofline uses (`roofline.kernel_ms`).

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
    "config5": (65536, (1200,), None, "configs[4]: 65536 streams x 1 s @1200 baud, clean, per GPU (524288 on 8)"),
    "config2": (4096, (1200,), None, "configs[1]: 4096 streams x 1 s @1200 baud, clean, per GPU"),
    "config3": (65536, (300, 1200, 2400), None, "configs[2]: 65536 streams x 1 s mixed baud {300,1200,2400}, clean, per GPU"),
    "config4": (65536, (1200,), 10.0, "configs[3]: 65536 streams x 1 s @1200 baud, additive noise SNR 10 dB, per GPU"),
    # not a BASELINE config: --workload custom --bauds 480,12000 [--streams N] times any baud mix
    # (profiles of the rates furthest from the roofline)
    "custom": (4096, (1200,), None, "custom: streams x 1 s, clean, bauds from --bauds, per GPU"),
    # r6: the config5 shard as a gate-cut capture would deliver it -- every stream starts with its own pseudo-random
    # lead-in of 0 ... 2047 samples of low-level noise (|x| < 600) before the Transmitter frame, so the clock index
    # (ref:322-339) is arbitrary: 7 of 8 streams have (2 * ci) & 15 != 0 and none takes the ci == 0 shortcut
    "config5_lead": (65536, (1200,), None, "config5 shard with a per-stream lead-in of 0..2047 noise samples (arbitrary clock index): 65536 streams x 1 s @1200 baud, per GPU"),
}
# workloads whose streams start with a lead-in: exclusive upper bound of the per-stream lead (samples); --lead N|random
# sets it for `custom` (N: the same lead for every stream)
LEADS = {"config5_lead": 2048}
LEAD_FIXED = {}                 # name -> fixed lead (samples) instead of a per-stream random one
LEAD_NOISE = 600                # |x| < this in the lead-in (the limiter's dead zone is 512, the squelch 14000)
LEAD_SEED = 777
HEADLINE = "config5"            # the headline workload at EVERY N (one weak-scaling curve)
NEXT_ROWS = ("f1_modulate", "f2_gate", "f3_wav_ingest", "f5_wav_egress", "f2_chain", "ragged_lengths")   # SURVEY 8(f) rows (+ the egress) as sub-records at N = 1
RATES_ROW = "rates_4096"        # 4096 x 1 s at EVERY rate a Receiver can be built for (36 values of bit_frames)
RATES_BIG_ROW = "rates_65536"   # the same at 65536 streams: the steady-state fraction per rate (no launch-shape quantisation)
RATES_ROWS = {RATES_ROW: 4096, RATES_BIG_ROW: 65536}
# what the default N = 1 run carries besides configs 2-4: the three SURVEY 8(f) rows and the steady-state per-rate
# table.  f5_wav_egress (not a SURVEY row) and rates_4096 (launch-shape quantisation, documented and closed) are
# measured on request: --sub f5_wav_egress,rates_4096
DEFAULT_RIDERS = ("f1_modulate", "f2_gate", "f2_chain", "f3_wav_ingest", RATES_BIG_ROW)
# 48000 / baud must divide 48000 and be a multiple of 4: 12000 ... 24 baud
ALL_RATES = tuple(48000 // bf for bf in range(4, 2048, 4) if 48000 % bf == 0)
BER_SNRS = (30, 25, 20, 15, 10, 7, 5, 3, 0)   # configs[3] sweep 30 -> 5 dB + two points below it
STREAM_LEN = 48000
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
MALL_BYTES = 256 << 20  # Infinity Cache: inputs smaller than a few of these are rotated
METRIC = "Msamples/s demodulated (batched 48 kHz streams) + decoded-byte match rate vs CPU ref"
PAYLOAD_SEED = 2024
RATE_ORDER = "cycle"            # --rate-order: how --bauds are laid over the streams of a custom workload
TRAINING_TIME = 0.5             # --training-time (custom workloads): the Transmitter's training_time (ref:437-438), default 0.5 s


def kernel_source_hash() -> str:
    """sha256 over the demod kernel's sources (afsk_demod*.h/.hip + afsk_kernels.h): ties a
    committed rocprof figure (profiles/traffic_latest.json) to the kernel it was measured on."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "afskmodem_amd", "csrc")
    for fn in sorted(os.listdir(d)):
        # (build.sh: the compiler flags are part of what a kernel is -- r5 added an -mllvm option worth 2 - 6 %)
        if (fn.endswith((".h", ".hip")) and (fn.startswith("afsk_demod") or fn == "afsk_kernels.h")) or fn == "build.sh":
            h.update(fn.encode())
            h.update(open(os.path.join(d, fn), "rb").read())
    return h.hexdigest()[:16]


def plan(world: int, workload: str = "", sub=None, streams: int = 0) -> dict:
    """What one run measures -- pure, no GPU (tests/test_bench_launch.py checks it on CPU).
    The headline is the SAME workload at every N; the defaults only differ in what rides along."""
    main = workload or HEADLINE
    if sub is not None:
        names = [x for x in sub.split(",") if x]
    elif workload or streams:
        names = []
    else:
        names = ["config2", "config3", "config4", "config5_lead"] + list(DEFAULT_RIDERS) if world == 1 else ["config2"]
    riders = NEXT_ROWS + tuple(RATES_ROWS)
    for x in [main] + names:
        if x not in WORKLOADS and x not in riders:
            raise SystemExit(f"unknown workload {x}")
    if main in riders:
        raise SystemExit("f1 / f2 / f3 / rates_* are sub-records, not headline workloads")
    return {"main": main, "subs": [x for x in names if x in WORKLOADS and x != main],
            "next": [x for x in names if x in riders]}


# --share-gpu0 (diagnostic): the GPU boxes kill a run with more than 6 processes on one device; a test
# runner that holds the device itself plus 4 ranks stays below that
SHARE_GPU0_MAX_RANKS = 4


def config_block(name: str, n_local: int, world: int, backend: str = "nccl", share_gpu0: bool = False) -> dict:
    """The `config` object of the result line (no GPU needed to build it)."""
    _, bauds, snr, desc = WORKLOADS[name]
    par = f"stream-sharded x{world}" + (" + RCCL gather of decoded records" if world > 1 else "")
    if backend != "nccl":
        par += " [DIAGNOSTIC: gloo backend" + (", all ranks on one GPU" if share_gpu0 else "") + "]"
    return {"workload": desc, "streams_per_gpu": n_local, "streams_total": n_local * world,
            "stream_len": STREAM_LEN, "bauds": list(bauds), "snr_db": snr, "parallelism": par}


# launcher, heartbeats, watchdog, the diagnostic line: tools/bench_launch.py (no torch, no HIP; every name re-exported)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_launch  # noqa: E402
from bench_launch import (EXIT_DEADLINE, EXIT_NO_LINE, EXIT_RANK_FAILED, Heartbeat, Watchdog, failure_line,  # noqa: E402,F401
                          free_port, hb_dir_for_env, read_heartbeats, read_partial, self_launch, usable_cpus, visible_gpus)


# --------------------------------------------------------------------------- one rank


class Ctx:
    """Per-process state: device, streams, process group."""

    def __init__(self, args, hb: "Heartbeat | None" = None):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if self.world != args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={self.world}")
        self.hb = hb or Heartbeat(self.rank, self.world)
        self.hb.beat("importing torch")
        import datetime
        import torch
        import torch.distributed as dist
        from afskmodem_amd import _native
        self.torch, self.dist, self.args = torch, dist, args
        self.hb.beat("checking for a HIP device")
        _native.require_device()          # no GPU -> loud failure, never a CPU fallback
        # --share-gpu0 (diagnostic, with --dist-backend gloo): every rank uses device 0, so the whole
        # N > 1 flow (launcher, ranks, gather, per-rank checks) can be exercised on a 1-GPU box;
        # RCCL itself refuses two ranks on one device
        dev_index = 0 if args.share_gpu0 else self.local_rank
        if not args.share_gpu0 and dev_index >= torch.cuda.device_count():
            raise SystemExit(f"bench.py: rank {self.rank} wants device {dev_index} but only "
                             f"{torch.cuda.device_count()} GPU(s) are visible")
        torch.cuda.set_device(dev_index)
        self.dev = torch.device("cuda", dev_index)
        # one process per GPU, on the socket its GPU hangs off (host side of the PCIe traffic: f3, host entries)
        from afskmodem_amd import dist as adist
        self.numa = adist.bind_to_device_numa_node(self.dev)
        self.hb.beat(f"device set: cuda:{dev_index} ({torch.cuda.get_device_name(dev_index)}), numa {self.numa}")
        self.use_dist = self.world > 1 or args.force_gather
        if self.use_dist:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if "MASTER_PORT" not in os.environ:
                os.environ["MASTER_PORT"] = str(free_port())   # single-rank --force-gather only
            # a rank that never arrives must not hold the others for torch's default 10 / 30 minutes
            tmo = datetime.timedelta(seconds=args.pg_timeout_s)
            if args.dist_backend == "nccl":
                dist.init_process_group("nccl", rank=self.rank, world_size=self.world, device_id=self.dev, timeout=tmo)
            else:
                dist.init_process_group(args.dist_backend, rank=self.rank, world_size=self.world, timeout=tmo)
            self.hb.beat(f"pg up: backend {args.dist_backend}, world {self.world}, timeout {args.pg_timeout_s:g} s")
        self.backend = args.dist_backend
        self.lib = _native.lib()
        self.cur = torch.cuda.current_stream()
        self.comm = torch.cuda.Stream(device=self.dev) if self.use_dist else None
        if self.use_dist:
            # the first collective builds the communicator (RCCL: topology detection, xGMI rings): do it here,
            # under its own heartbeat, not inside the first timed fence
            one = torch.ones(1, dtype=torch.int32, device=self.dev)
            self.all_reduce(one)
            torch.cuda.synchronize()
            self.hb.beat(f"first collective done: {int(one.item())} ranks answered")

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

    def collect(self, out, inp) -> None:
        """The exchange of the decoded records.  --gather-mode root (default): every rank's buffer goes to
        rank 0 (north_star: RCCL 'only to gather decoded byte buffers'; on the point-to-point xGMI fabric
        that is one direct transfer per rank into rank 0's seven ingress links, never a ring);
        --gather-mode all: all_gather_into_tensor, every rank ends up with every rank's records.
        `out` ([world * n] on every rank; only rank 0's is filled in root mode), `inp` [n]."""
        if self.args.gather_mode == "all":
            self.all_gather(out, inp)
            return
        n = inp.numel()
        if self.backend == "nccl":
            parts = [out[r * n: (r + 1) * n] for r in range(self.world)] if self.rank == 0 else None
            self.dist.gather(inp, parts, dst=0)
            return
        h_in = inp.cpu()
        h_parts = [self.torch.empty_like(h_in) for _ in range(self.world)] if self.rank == 0 else None
        self.dist.gather(h_in, h_parts, dst=0)
        if self.rank == 0:
            out.copy_(self.torch.cat(h_parts))

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

    def __init__(self, ctx: Ctx, name: str, n_local: int, snr_db=None, copies: int = 0, bauds=None, desc=None):
        from afskmodem_amd import batch, synth
        from afskmodem_amd import dist as adist
        torch = ctx.torch
        _, wl_bauds, wl_snr, wl_desc = WORKLOADS[name]
        bauds = tuple(bauds) if bauds is not None else wl_bauds
        desc = desc or wl_desc
        self.ctx, self.name, self.desc, self.bauds = ctx, name, desc, bauds
        self.snr_db = wl_snr if snr_db is None else snr_db
        self.n_local = n_local
        self.n_total = n_local * ctx.world
        self.first = ctx.rank * n_local
        assert adist.shard_range(self.n_total, ctx.rank, ctx.world) == (self.first, self.first + n_local)
        self.bf_h, self.plen_h, self.payload_h, ts_h = self.host_meta(self.first, n_local, bauds)
        self.off, self.ln = batch.uniform_layout(n_local, STREAM_LEN, ctx.dev)
        # lead-in workloads: the frame of stream s starts lead[s] samples into its slot (and loses as many samples of
        # its 4800-sample silent tail); the demodulator still gets the whole slot
        self.lead_h = self.host_leads(name, self.first, n_local)
        if self.lead_h is not None:
            lead_d = ctx.t(self.lead_h.astype(np.int64))
            self._moff, self._mln = self.off + lead_d, (self.ln.to(torch.int64) - lead_d).to(torch.int32)
        else:
            self._moff, self._mln = self.off, self.ln
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
        # one baud rate in the whole shard -> the Receiver-shaped uniform entry (bit_frames by value); several ->
        # the rate-grouped dispatch (the host knows every stream's rate: one launch over the rate-sorted streams);
        # --entry mixed forces the per-stream kernel (bit_frames[] in device memory) for A/B runs
        self.uniform_bf = int(self.bf_h[0]) if (len(set(bauds)) == 1 and ctx.args.entry != "mixed") else None
        self.plan = None
        if self.uniform_bf is None and ctx.args.entry != "mixed":
            self.plan = batch.GroupPlan(self.bf_h, ctx.dev)
        self.stride = batch.out_stride_for(STREAM_LEN, int(self.bf_h.min()))
        _, self.flat_sz = batch.flat_layout(n_local, self.stride)
        torch.cuda.synchronize()

    @staticmethod
    def host_meta(first: int, n: int, bauds):
        from afskmodem_amd import synth
        gidx = np.arange(first, first + n)
        if RATE_ORDER == "blocks":      # (A/B) rates in contiguous blocks of streams instead of cycling per stream
            baud_arr = np.asarray([bauds[min(int(i - first) * len(bauds) // max(n, 1), len(bauds) - 1)] for i in gidx], np.int32)
        else:
            baud_arr = np.asarray([bauds[i % len(bauds)] for i in gidx], np.int32)
        bf_h = (48000 // baud_arr).astype(np.int32)
        plen_h = np.asarray([synth.one_second_payload(int(b), TRAINING_TIME) for b in baud_arr], np.int32)
        pstride = max(synth.one_second_payload(int(b), TRAINING_TIME) for b in bauds)
        payload_h = synth.payload_bytes(PAYLOAD_SEED, first, n, pstride)
        ts_h = np.asarray([synth.ts_cycles_for(int(b), TRAINING_TIME) for b in baud_arr], np.int32)
        return bf_h, plen_h, payload_h, ts_h

    @staticmethod
    def host_leads(name: str, first: int, n: int):
        """Per-stream lead-in (samples) of a lead-in workload, a pure function of the GLOBAL stream index; None otherwise."""
        if name in LEAD_FIXED:
            return np.full(n, LEAD_FIXED[name], np.int32)
        if name not in LEADS:
            return None
        from afskmodem_amd import synth
        g = np.arange(first, first + n, dtype=np.uint64)
        return (synth._hash32((g * np.uint64(2654435761) + np.uint64(LEAD_SEED)) & np.uint64(0xFFFFFFFF)) % np.uint64(LEADS[name])).astype(np.int32)

    def write_leads(self) -> None:
        """Low-level integer noise (|x| < LEAD_NOISE, a hash of global stream index and position) over the lead-in."""
        torch = self.ctx.torch
        x2 = self.inputs[0].view(self.n_local, STREAM_LEN)
        width = int(self.lead_h.max()) if self.lead_h.size else 0
        if width <= 0:
            return
        lead_d = self.ctx.t(self.lead_h.astype(np.int64))
        for s0 in range(0, self.n_local, 8192):          # (bounded temporaries: 8192 x 2048 int64)
            s1 = min(self.n_local, s0 + 8192)
            g = torch.arange(self.first + s0, self.first + s1, device=self.ctx.dev, dtype=torch.int64)[:, None]
            j = torch.arange(width, device=self.ctx.dev, dtype=torch.int64)[None, :]
            h = (g * 2048 + j) * 0x9E3779B1 + LEAD_SEED
            h = (h ^ (h >> 15)) * 0x85EBCA6B & 0xFFFFFFFF
            h = (h ^ (h >> 13)) & 0xFFFFFFFF
            noise = (h % (2 * LEAD_NOISE - 1) - (LEAD_NOISE - 1)).to(torch.int16)
            head = x2[s0:s1, :width]
            head.copy_(torch.where(j < lead_d[s0:s1, None], noise, head))

    def regenerate(self, snr_db, seed: int = 99) -> None:
        """(Re)write inputs[0]: Transmitter frames (+ the lead-in, + additive noise at snr_db)."""
        from afskmodem_amd import batch, synth
        x = self.inputs[0]
        # the .wav writer's decimate/duplicate quirk (ref:239-244) is part of every Transmitter.save
        # stream; at 12000 baud it destroys the mark tone (in the reference too), so a custom
        # workload with that rate uses the ideal frames
        batch.modulate_batch(self._payload_d, self._plen_d, self.bf, self._ts_d, self._moff, self._mln,
                             STREAM_LEN, x, 12000 not in self.bauds)
        if self.lead_h is not None:
            self.write_leads()
        if snr_db is not None:
            batch.add_noise_batch(x, self.off, self.ln, STREAM_LEN, synth.snr_to_scale_q24(snr_db),
                                  seed=seed, stream_idx_base=self.first)
        self.ctx.torch.cuda.synchronize()


def weighted_sum(torch, flat):
    """Order-sensitive checksum of a uint8 tensor (int64 arithmetic on the device)."""
    w = (torch.arange(flat.numel(), device=flat.device, dtype=torch.int64) % 65521) + 1
    return (flat.to(torch.int64) * w).sum()


def median(v):
    v = sorted(v)
    return v[len(v) // 2] if v else None


def measure(ctx: Ctx, sh: Shard, steps: int, warmup: int, preroll_ms: float, gather_every: int = 0,
            min_region_ms: float = 50.0, max_regions: int = 64, with_gather: bool = True):
    """Pre-roll, W warm-up steps, then timed regions of EXACTLY K steps between two fences each: one
    region, or -- when a region is shorter than min_region_ms -- as many as add up to it; the reported
    region is the median one.  Every step writes its own output slot; with a process group the slots
    of G consecutive steps are gathered by ONE collective on a side stream while the next group
    runs.  with_gather=False: the same regions (fences and the max over ranks included) without the
    exchange of the decoded records -- `value_no_gather`, so that weak-scaling efficiency and the cost of
    the gather separate."""
    from afskmodem_amd import _native, batch
    torch, dist = ctx.torch, ctx.dist
    cur, comm = ctx.cur, (ctx.comm if with_gather else None)
    n_local, stride, flat_sz = sh.n_local, sh.stride, sh.flat_sz
    world, rank = ctx.world, ctx.rank

    # Each collective costs the compute stream ~40 us (cross-stream events around it), so a group
    # should cover a few ms of kernels -- but the LAST group's exchange has nothing left to overlap with and
    # sits inside the timed region, so groups stay small: ~4 ms = 64 steps of config #2, 3 of the 6.29 GB
    # ones (7.1 MB of records per rank and step there).
    est_step_s = 2.0 * n_local * STREAM_LEN / 6.0e12
    if gather_every > 0:
        G = gather_every
    else:
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
    nin = len(sh.inputs)
    # argument tuples are built once: the timed loop is one ctypes call per step
    if sh.uniform_bf is not None:
        fn = ctx.lib.afsk_demod_batch_uniform
        slot_args = [[(x.data_ptr(), sh.off.data_ptr(), sh.ln.data_ptr(), sh.uniform_bf, 14000, n_local,
                       o.bytes.data_ptr(), stride, o.nbytes.data_ptr(), o.nbits.data_ptr(),
                       o.clock_idx.data_ptr(), o.term_frame.data_ptr(), o.status.data_ptr(), None, None, 0, sptr)
                      for o in slots] for x in sh.inputs]
    elif sh.plan is not None:
        fn = ctx.lib.afsk_demod_batch_grouped
        slot_args = [[(sh.plan.handle, x.data_ptr(), sh.off.data_ptr(), sh.ln.data_ptr(), 14000,
                       o.bytes.data_ptr(), stride, o.nbytes.data_ptr(), o.nbits.data_ptr(),
                       o.clock_idx.data_ptr(), o.term_frame.data_ptr(), o.status.data_ptr(), None, None, 0, sptr)
                      for o in slots] for x in sh.inputs]
    else:
        fn = ctx.lib.afsk_demod_batch
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
            ctx.collect(gath_bufs[g & 1], out_all[gr * G * flat_sz: (gr + 1) * G * flat_sz])
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

    # HIP events inside the region: one per `eg` steps (every step when a step is >= ~200 us), so the
    # records themselves cannot open gaps between 60 us kernels
    eg = max(1, min(steps, int(-(-200e-6 // est_step_s)))) if steps > 0 else 1
    region_s: list = []          # wall clock of each region (max over ranks)
    region_event_ms: list = []   # HIP-event time of each region on this rank
    interval_ms: list = []       # HIP-event time per step of every eg-step interval
    host_issue_s = 0.0
    n_regions = 1
    r = 0
    while r < n_regions:
        out_all.zero_()            # a correct slot after a timed region was written BY that region
        for gr in range(ngroups):
            gathered_once[gr] = False
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps // eg + 2)]
        ctx.fence()
        state["timing"] = True
        t0 = time.perf_counter()
        marks[0].record(cur)
        nm = 1
        for i in range(steps):
            step(i)
            if (i + 1) % eg == 0 and i + 1 < steps:
                marks[nm].record(cur)
                nm += 1
        finish(steps)
        marks[nm].record(cur)
        host_issue_s = time.perf_counter() - t0        # host time to enqueue the whole timed region
        ctx.fence()
        elapsed = time.perf_counter() - t0
        state["timing"] = False
        if ctx.use_dist:
            tmax = torch.tensor([elapsed], dtype=torch.float64, device=ctx.dev)
            ctx.all_reduce(tmax, op=dist.ReduceOp.MAX)
            elapsed = float(tmax.item())              # identical on every rank from here on
        region_s.append(elapsed)
        region_event_ms.append(marks[0].elapsed_time(marks[nm]))
        for k in range(nm):
            n_in = eg if k + 1 < nm else steps - eg * (nm - 1)
            if n_in > 0:
                interval_ms.append(marks[k].elapsed_time(marks[k + 1]) / n_in)
        if r == 0 and steps > 0 and min_region_ms > 0:
            n_regions = max(1, min(max_regions, int(-(-min_region_ms * 1e-3 // max(elapsed, 1e-6)))))
        r += 1
    elapsed = median(region_s)
    # The roofline's launch duration comes from the SAME region the value is quoted on -- the one with the median
    # wall clock -- as the average of its HIP-event time over its K launches (gaps included).  (r5: it used to be the
    # average over all regions, and one 7 ms hiccup of the box in one of three regions -- value unaffected -- put a
    # rate of the per-rate table at 0.595 beside a per-launch median of 0.852.)
    mid = min(range(n_regions), key=lambda k: (abs(region_s[k] - elapsed), k))
    event_ms = sum(region_event_ms)
    kernel_ms = region_event_ms[mid] / max(steps, 1)

    # ---- which step gets the full check: a random one among those whose slot still holds it (last region)
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
    ev_med = median(interval_ms)
    rec = {
        "workload": sh.desc,
        "streams_per_gpu": n_local,
        "entry": ("afsk_demod_batch_uniform" if sh.uniform_bf is not None else
                  "afsk_demod_batch_grouped" if sh.plan is not None else "afsk_demod_batch"),
        "rates_in_batch": 1 if sh.plan is None else len(sh.plan.groups()),
        "steps": steps, "warmup": warmup, "preroll_launches": preroll_launches,
        "value": round(samples_per_step * steps / elapsed / 1e6, 1),            # wall clock of the median region, fences included
        "value_event_time": round(samples_per_step * steps * n_regions / (event_ms * 1e-3) / 1e6, 1),   # HIP events, all regions
        # the same wall-clock rate counting only the samples the reference has to read (up to the
        # squelch-triggering symbol); `value` counts every sample of the buffers, tail silence included
        "value_active_samples": round(float(active.sum()) * ctx.world * steps / elapsed / 1e6, 1),
        "unit": "Msamples/s",
        "ms_per_step": round(elapsed / max(steps, 1) * 1e3, 5),
        "timed_regions": n_regions,
        "timed_region_ms": round(elapsed * 1e3, 4),
        "timed_region_ms_all": [round(x * 1e3, 4) for x in region_s],
        "timed_ms_total": round(sum(region_s) * 1e3, 3),
        "event_ms_per_step_median": None if ev_med is None else round(ev_med, 5),
        "event_intervals": {"count": len(interval_ms), "steps_per_interval": eg,
                            "min_ms_per_step": round(min(interval_ms), 5) if interval_ms else None,
                            "max_ms_per_step": round(max(interval_ms), 5) if interval_ms else None},
        "roofline": {"bound": "hbm", "achieved": round(achieved_gbs, 1), "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": round(achieved_gbs / HBM_PEAK_GBS, 4),
                     "traffic": None,
                     "algorithmic_bytes_per_launch": alg_bytes,
                     "kernel_ms": round(kernel_ms, 5),
                     "kernel_ms_median": None if ev_med is None else round(ev_med, 5),
                     "frac_at_median": None if ev_med is None else round(alg_bytes / (ev_med * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                     "full_buffer_gbs": round((2 * n_local * STREAM_LEN) / (kernel_ms * 1e-3) / 1e9, 1)},
        "input_buffers_rotated": nin,
        # streams whose clock index (ref:322-339) is not a multiple of 8 samples / is not zero: what a capture that does
        # not start on the training sequence's first sample looks like (r6; 0.0 / 0.0 for every Transmitter-shaped batch)
        "clock_index_unaligned_share": round(float(((res.clock_idx.astype(np.int64) * 2) & 15).astype(bool).mean()), 4),
        "clock_index_nonzero_share": round(float((res.clock_idx != 0).mean()), 4),
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
        have_all = ctx.args.gather_mode == "all"
        for rr in (range(world) if (have_all or rank == 0) else ()):
            part = gbuf[rr * G * flat_sz: (rr + 1) * G * flat_sz]
            ok_sum = bool((weighted_sum(torch, part) == sums[rr]).item())
            v = batch.views_of_flat(part[k_last * flat_sz: (k_last + 1) * flat_sz], n_local, stride)
            ok_pay = None
            if sh.snr_db is None:
                _, plen_r, payload_r, _ = Shard.host_meta(rr * n_local, n_local, sh.bauds)
                pl = ctx.t(plen_r)
                exp = ctx.t(payload_r)
                width = min(int(exp.shape[1]), stride)
                col = torch.arange(width, device=ctx.dev)[None, :]
                m = col < pl[:, None]
                same = ((v.bytes[:, :width] == exp[:, :width]) | ~m).all(dim=1) & (v.nbytes == pl)
                ok_pay = bool(same.all().item())
            per_rank.append(ok_sum and ok_pay is not False)
        # root mode: rank 0 holds and checks every rank's slice (the other ranks contribute a neutral 1);
        # all mode: every rank checks every slice
        mine = torch.tensor([int(all(per_rank))], dtype=torch.int32, device=ctx.dev)
        ctx.all_reduce(mine, op=dist.ReduceOp.MIN)
        ones = torch.ones(1, dtype=torch.int32, device=ctx.dev)
        ctx.all_reduce(ones, op=dist.ReduceOp.SUM)
        rec["ranks_seen"] = int(ones.item())
        rec["gather_mode"] = "all_gather_into_tensor (every rank)" if have_all else "gather to rank 0"
        rec["gather_check"] = per_rank                       # rank 0's view: one entry per producer rank
        rec["gather_check_on_every_rank"] = bool(mine.item())   # (root mode: rank 0's verdict, agreed by all_reduce)
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
        res = batch.demod_batch(sh.inputs[0], sh.off, sh.ln, sh.uniform_bf if sh.uniform_bf is not None else sh.bf, 14000, out_stride=sh.stride)
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
    # (the line says where the figure comes from: a committed PMC pass of the same kernel source, not this run)
    rf["traffic_source"] = f"profiles/traffic_latest.json@{src_hash} (rocprofv3 --pmc passes of {ent.get('source')}; not measured in this run)"


def cpu_calibration() -> dict | None:
    """{file, pyref_over_reference (1200 baud), range} from the committed calibration of oracle/pyref.py against the
    real reference (build container: tools/calibrate_cpu_reference.py); None when the file is absent."""
    rel = os.path.join("profiles", "cpu_reference_calibration.json")
    try:
        cj = json.load(open(os.path.join(ROOT, rel)))
        return {"file": rel, "pyref_over_reference": cj["pyref_over_reference_1200"],
                "range": [cj["pyref_over_reference_min"], cj["pyref_over_reference_max"]],
                "reference_msamples_per_s_1200_build_container": cj["by_baud"]["1200"]["reference_msamples_per_s"]}
    except Exception:  # noqa: BLE001
        return None


def cpu_baseline_for(sh: Shard, res, got_payloads, ns: int, cores: int, budget_s: float = 10.0, idx=None, label=""):
    """The CPU oracle (C port of the reference hot path) timed on this box's host cores on a bounded
    sample of the shard (the first ns streams, or the streams listed in idx), checked against the GPU's
    outputs first; plus the pure-Python 'reference-shaped' restatement on one core."""
    from oracle import afsk_oracle as O   # checker + reported CPU baseline only
    from oracle import pyref
    if idx is None:
        idx = np.arange(ns)
    idx = np.asarray(idx[:ns], dtype=np.int64)
    ns = int(idx.size)
    x = sh.inputs[0].view(sh.n_local, STREAM_LEN)
    h = x[sh.ctx.torch.from_numpy(idx).to(x.device)].cpu().numpy().reshape(-1)
    h_off = np.arange(ns, dtype=np.int64) * STREAM_LEN
    h_ln = np.full(ns, STREAM_LEN, np.int32)
    bf = np.ascontiguousarray(sh.bf_h[idx])
    want = O.demod_batch(h, h_off, h_ln, bf, 14000, out_stride=sh.stride, n_threads=cores)
    match = 0
    for j, s_i in enumerate(idx):
        nb = int(want["nbytes"][j])
        m = min(nb, sh.stride)
        match += bool(nb == int(res.nbytes[s_i]) and int(want["nbits"][j]) == int(res.nbits[s_i])
                      and int(want["clock_idx"][j]) == int(res.clock_idx[s_i])
                      and int(want["term_frame"][j]) == int(res.term_frame[s_i])
                      and want["bytes"][j, :m].tobytes() == got_payloads[s_i][:m])
    n1 = max(ns // 8, 1)
    t1 = time.perf_counter()
    O.demod_batch(h[: n1 * STREAM_LEN], h_off[:n1], h_ln[:n1], bf[:n1], 14000, out_stride=sh.stride, n_threads=1)
    dt1 = time.perf_counter() - t1
    reps = 0
    t2 = time.perf_counter()
    while True:
        O.demod_batch(h, h_off, h_ln, bf, 14000, out_stride=sh.stride, n_threads=cores)
        reps += 1
        if time.perf_counter() - t2 > budget_s or reps >= 50:
            break
    dtc = (time.perf_counter() - t2) / reps
    # "reference-shaped" figure: the pure-Python restatement (oracle/pyref.py) on one core;
    # in the dev container it runs at 1.01-1.07x the speed of the real afskmodem.py.
    npy = min(3, ns)
    t3 = time.perf_counter()
    for j in range(npy):
        data, _, _, _ = pyref.demod(h[j * STREAM_LEN: (j + 1) * STREAM_LEN].tolist(), int(bf[j]))
        assert data == got_payloads[int(idx[j])], "pure-Python restatement disagrees with the GPU"
    dtp = (time.perf_counter() - t3) / npy
    doc = {
        "value": round(ns * STREAM_LEN / dtc / 1e6, 1), "unit": "Msamples/s", "cores": cores,
        "host_cores": os.cpu_count(),      # what the host shows; `cores` = usable_cpus() = the threads used (cgroup quota)
        "kind": "port",
        "sample": f"{ns} streams of the same batch{label}, CPU oracle (C port of afskmodem.py hot path), "
                  f"{cores} threads, {reps} reps; single thread on {n1} streams: "
                  f"{round(n1 * STREAM_LEN / dt1 / 1e6, 1)} Msamples/s",
        "single_thread_value": round(n1 * STREAM_LEN / dt1 / 1e6, 1),
        "python_reference_shaped_value": round(STREAM_LEN / dtp / 1e6, 3),
        "python_reference_shaped_note": f"oracle/pyref.py (pure-Python restatement, CPython, 1 core, {npy} streams); "
                                        "its speed relative to the real afskmodem.py, measured in the build container on "
                                        "the same streams, is in `calibration` (profiles/cpu_reference_calibration.json, "
                                        "written by tools/calibrate_cpu_reference.py); the reference itself never travels",
    }
    cal = cpu_calibration()
    if cal:
        doc["calibration"] = cal
    return doc, match / ns, ns


LINE_CAP = 4096   # bytes: the driver keeps ~8 KB of stdout + stderr; the result line must fit well inside

# keys of the full record that the one-line result keeps verbatim (the contract's keys first)
_LINE_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "entry", "timed_regions", "timed_region_ms",
              "event_ms_per_step_median", "kernel_source_hash", "roundtrip_match_rate", "all_timed_steps_identical",
              "match_rate", "match_sample_streams", "per_workload_value", "ranks_seen", "gather_mode", "gather_check",
              "gather_check_on_every_rank", "gather_every_steps", "gathers_in_timed_region",
              "value_no_gather", "ms_per_step_no_gather", "frac_no_gather", "host_threads_per_rank")
_ROOF_KEYS = ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "algorithmic_bytes_per_launch",
              "kernel_ms")
_CPU_KEYS = ("value", "unit", "cores", "kind", "sample", "single_thread_value", "python_reference_shaped_value",
             "calibration")


def _pick(d: dict, keys) -> dict:
    return {k: d[k] for k in keys if k in d}


def _sub_summary(name: str, rec: dict) -> dict:
    """One sub-record reduced to {value, frac, match_rate} (+ the two or three figures that only it has)."""
    if "error" in rec and "roofline" not in rec:
        return {"error": str(rec["error"])[:120]}                        # a rider that failed: said so, nothing else
    if "roofline" not in rec and all(isinstance(v, dict) for v in rec.values()):
        return {k: _sub_summary(k, v) for k, v in rec.items()}          # f2_gate: one entry per capture count
    if "by_baud" in rec:                                                 # rates_4096 / rates_65536
        out = _pick(rec, ("min_frac", "max_frac", "median_frac", "rates_below_0.60", "rates_below_0.75",
                          "slowest", "all_round_trips_exact", "min_match_rate"))
        for k in ("rates_below_0.60", "rates_below_0.75"):               # a long list becomes its length
            if isinstance(out.get(k), list) and len(out[k]) > 6:
                out[k] = len(out[k])
        return out
    out = {"value": rec.get("value"), "frac": (rec.get("roofline") or {}).get("frac")}
    for k in ("match_rate", "oracle_match_rate", "decoded_match_rate"):
        if k in rec:
            out["match_rate"] = rec[k]
    if "roundtrip_match_rate" in rec and rec["roundtrip_match_rate"] is not None:
        out["roundtrip"] = rec["roundtrip_match_rate"]
    for k in ("chain_ms", "ragged_over_uniform_time_per_byte_planned", "ragged_over_uniform_time_per_byte_stream_order"):
        if k in rec:
            out[{"chain_ms": "chain_ms", "ragged_over_uniform_time_per_byte_planned": "ragged_vs_1s", "ragged_over_uniform_time_per_byte_stream_order": "ragged_vs_1s_unplanned"}[k]] = rec[k]
    if "frac_overwrite_in_place" in (rec.get("roofline") or {}):
        out["frac_overwrite"] = rec["roofline"]["frac_overwrite_in_place"]
    if "entry" in rec:
        out["entry"] = rec["entry"].replace("afsk_demod_batch", "demod")
    if rec.get("clock_index_unaligned_share"):
        out["ci_unaligned"] = rec["clock_index_unaligned_share"]
    if "ber_curve" in rec:
        rows = rec["ber_curve"]
        out["ber"] = {str(r["snr_db"]): r["ber_gpu_all_streams"] for r in rows}
        out["ber_equals_cpu"] = all(r["ber_equal"] and r["cpu_match_rate"] == 1.0 for r in rows)
    if "gather_ms" in rec:
        out["gather_ms"] = rec["gather_ms"]["median"]
        out["gather_check"] = rec.get("gather_check_on_every_rank")
    return out


def compact_line(full: dict, full_path: str | None = None) -> dict:
    """The ONE line the driver parses: the contract's keys, `roofline`, `cpu_baseline`, the match rates and a
    three-figure summary per sub-record -- under LINE_CAP bytes whatever rides along.  Everything else (per-rate
    table, BER rows, event intervals, notes) stays in the full record (`full_record`, a file next to the run)."""
    line = _pick(full, _LINE_KEYS)
    line["roofline"] = _pick(full["roofline"], _ROOF_KEYS)
    if "cpu_baseline" in full:
        cb = _pick(full["cpu_baseline"], _CPU_KEYS)
        if len(cb.get("sample", "")) > 160:
            cb["sample"] = cb["sample"][:157] + "..."
        line["cpu_baseline"] = cb
    if "gather_ms" in full:
        line["gather_ms"] = _pick(full["gather_ms"], ("median", "max", "bytes_per_rank"))
    if "headline_again_at_end" in full:
        line["headline_again_at_end"] = _pick(full["headline_again_at_end"], ("value", "frac", "error"))
    if full.get("sub_records"):
        line["sub_records"] = {k: _sub_summary(k, v) for k, v in full["sub_records"].items()}
    line["full_record"] = full_path
    # belt and braces: should a future rider push the line over the cap, drop the summaries, never the contract
    for victim in ("sub_records", "per_workload_value", "gather_check"):
        if len(json.dumps(line)) < LINE_CAP:
            break
        line[victim] = "see full_record"
    if len(json.dumps(line)) >= LINE_CAP:
        # last resort (never an exception after a multi-minute run): the contract's keys, the roofline and the
        # pointer to the full record -- a parseable line is always printed
        keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "value_no_gather", "ranks_seen")
        slim = _pick(line, keep)
        slim["config"] = _pick(line.get("config") or {}, ("workload", "streams_per_gpu", "streams_total", "parallelism"))
        slim["roofline"] = _pick(line.get("roofline") or {}, ("bound", "achieved", "peak", "unit", "frac", "traffic"))
        if "cpu_baseline" in line:
            slim["cpu_baseline"] = _pick(line["cpu_baseline"], ("value", "unit", "cores", "kind"))
            slim["cpu_baseline"]["sample"] = "see full_record"
        slim["full_record"] = full_path
        slim["line_note"] = "summaries dropped: the compact line would have exceeded the cap"
        line = slim
    return line


def write_full_record(full: dict, world: int, tag: str = "") -> str | None:
    """The full record (tens of KB) goes to gpurun_out/bench_full_n<N>.json (the default run; an explicit
    --workload adds its name, so side runs do not overwrite the headline's record) -- never to stdout or
    stderr, whose tails are all the driver keeps."""
    for d in (os.path.join(ROOT, "gpurun_out"), ROOT):
        try:
            os.makedirs(d, exist_ok=True)
            path = os.path.join(d, f"bench_full_n{world}{'_' + tag if tag else ''}.json")
            with open(path, "w") as f:
                json.dump(full, f, indent=1)
            return os.path.relpath(path, ROOT)
        except OSError:
            continue
    return None


bench_launch.configure(METRIC, WORKLOADS[HEADLINE][3], LINE_CAP)


def run_rank(args) -> None:
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    hb = Heartbeat(rank, world)
    deadline_s = args.deadline_s
    if os.environ.get("AFSK_BENCH_LAUNCHER") == "1" and deadline_s > 0:
        deadline_s += 15.0            # bench.py's own launcher holds the clock; this is only the backstop behind it
    dog = Watchdog(hb, args, deadline_s)
    try:
        _run_rank(args, hb, dog)
    except BaseException as e:  # noqa: BLE001
        if isinstance(e, SystemExit) and e.code in (0, None):
            raise
        # an ordinary Python error on this rank: say so in the line (rank 0) or on stderr, exit non-zero --
        # torchrun / the launcher then terminates the other ranks, whose watchdogs stand by
        import traceback
        tb = traceback.format_exc()
        sys.stderr.write(tb)
        why = f"{type(e).__name__}: {e}"
        last = hb.phase
        hb.quiet = True
        hb.beat(f"failed in '{last}': {why}"[:200])
        dog.stand_down()
        if not dog.under_launcher:
            # through the same claim file as the watchdog (it may be printing at this very moment: torchrun's SIGTERM
            # and the exception the dead peer causes in a collective can arrive together): ONE line per job.  Rank 0
            # claims at once, another rank only if nobody has after its grace period.
            dog._maybe_print(EXIT_RANK_FAILED, why, last)
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(EXIT_RANK_FAILED if not isinstance(e, SystemExit) else (e.code if isinstance(e.code, int) else 2))


def _run_rank(args, hb: Heartbeat, dog: Watchdog) -> None:
    ctx = Ctx(args, hb)
    torch, dist = ctx.torch, ctx.dist
    world, rank = ctx.world, ctx.rank
    cores = usable_cpus()               # this rank's share of the box (LOCAL_WORLD_SIZE ranks per node)
    os.environ.setdefault("AFSK_IO_THREADS", str(min(32, cores)))     # libafsk_amd.so's I/O pool, same share
    os.environ.setdefault("AFSK_COPY_THREADS", str(min(16, cores)))
    src_hash = kernel_source_hash()

    pl = plan(world, args.workload, args.sub, args.streams)
    main_name, sub_names, next_rows = pl["main"], pl["subs"], pl["next"]

    def n_for(name: str) -> int:
        return args.streams if args.streams > 0 else WORKLOADS[name][0]

    # ---------------- main record
    hb.beat(f"building the {main_name} shard ({n_for(main_name)} streams)")
    sh = Shard(ctx, main_name, n_for(main_name))
    hb.beat("shard built")
    rec_ng = None
    if ctx.comm is not None and not args.no_value_no_gather:
        # N > 1: the SAME K-step regions first without the exchange of the decoded records ...
        hb.beat("timing without the gather")
        rec_ng, _ = measure(ctx, sh, args.steps, args.warmup, args.preroll_ms, args.gather_every, args.min_region_ms,
                            with_gather=False)
        hb.beat("timed without the gather")
    # ... then (the headline) with it: every step's records are exchanged inside the timed region
    hb.beat("warm-up + timed region")
    rec, aux = measure(ctx, sh, args.steps, args.warmup, args.preroll_ms if rec_ng is None else min(args.preroll_ms, 50.0),
                       args.gather_every, args.min_region_ms)
    hb.beat("timed" + (" and gathered" if ctx.comm is not None else ""))
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
        "config": config_block(main_name, sh.n_local, world, ctx.backend, args.share_gpu0),
        "entry": rec["entry"],
        "timed_regions": rec["timed_regions"],
        "timed_region_ms": rec["timed_region_ms"],
        "timed_region_ms_all": rec["timed_region_ms_all"],
        "timed_ms_total": rec["timed_ms_total"],
        "event_ms_per_step_median": rec["event_ms_per_step_median"],
        "event_intervals": rec["event_intervals"],
        "value_event_time": rec["value_event_time"],
        "value_active_samples": rec["value_active_samples"],
        "roofline": rec["roofline"],
        "kernel_source_hash": src_hash,
        "input_buffers_rotated": rec["input_buffers_rotated"],
        "roundtrip_match_rate": rec["roundtrip_match_rate"],
        "checked_step": rec["checked_step"],
        "all_timed_steps_identical": rec["all_timed_steps_identical"],
        "host_issue_ms_per_step": rec["host_issue_ms_per_step"],
        "host_threads_per_rank": cores,
    }
    out["config"]["bauds"] = list(sh.bauds)       # (--workload custom --bauds ... replaces the table entry)
    for k in ("ranks_seen", "gather_mode", "gather_check", "gather_check_on_every_rank", "gather_every_steps",
              "gathers_in_timed_region", "gather_ms"):
        if k in rec:
            out[k] = rec[k]
    if rec_ng is not None:
        out["value_no_gather"] = rec_ng["value"]
        out["ms_per_step_no_gather"] = rec_ng["ms_per_step"]
        out["frac_no_gather"] = rec_ng["roofline"]["frac"]
        out["roundtrip_match_rate_no_gather"] = rec_ng["roundtrip_match_rate"]
    full_path = [None]
    tag = args.workload or ("partial" if (args.sub is not None or args.streams) else "")

    def checkpoint(note: str) -> None:
        """Rank 0: from here on a line with numbers exists whatever happens to the riders -- the launcher or the
        watchdog prints this snapshot (marked incomplete) if the run dies or runs out of time later."""
        if rank != 0:
            return
        snap = dict(out)
        if subs:
            snap["sub_records"] = dict(subs)
        snap["per_workload_value"] = dict(per_workload)
        line = compact_line(snap, full_path[0])
        line["riders_pending"] = note
        hb.write_partial(line)

    subs: dict = {}
    per_workload = {main_name: out["value"]}
    checkpoint("everything after the headline")

    def rider(name: str, fn):
        """A sub-record that fails (or its import) becomes {"error": ...}: it can never cost the headline."""
        hb.beat(f"sub-record {name}")
        try:
            return fn()
        except Exception as e:  # noqa: BLE001
            import traceback
            sys.stderr.write(f"bench.py: sub-record {name} failed:\n{traceback.format_exc()}")
            if ctx.use_dist:
                raise             # at N > 1 a rank that skips a rider's collectives would hang the others: fail the run
            try:
                torch.cuda.synchronize()
            except Exception:  # noqa: BLE001
                pass
            return {"error": f"{type(e).__name__}: {e}"[:300]}

    if world == 1 and not args.no_cpu_baseline:
        hb.beat("cpu_baseline (CPU oracle on the host cores)")
        ns = args.cpu_sample_streams or min(sh.n_local, 4096)
        out["cpu_baseline"], out["match_rate"], out["match_sample_streams"] = cpu_baseline_for(
            sh, aux["res"], aux["got_payloads"], ns, cores, args.cpu_budget_s, label=" (the first of them)")
        checkpoint("every sub-record")
    # the next rows that reuse the headline shard's buffers: f1 re-writes its input, f2 reads it
    if world == 1 and ("f1_modulate" in next_rows or "f2_gate" in next_rows):
        def _rows():
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import bench_rows
            return bench_rows
        if "f1_modulate" in next_rows:
            subs["f1_modulate"] = rider("f1_modulate", lambda: _rows().measure_modulate(ctx, sh, args.next_reps))
        if "f2_gate" in next_rows:
            subs["f2_gate"] = {"captures_%d" % sh.n_local: rider("f2_gate", lambda: _rows().measure_gate(ctx, sh, args.next_reps))}
    del sh, aux            # (back to torch's caching allocator, NOT to the driver: see bench_rows.measure_rates)

    # ---------------- sub-records: the other single-GPU configs in the same line
    def config_sub(name: str) -> dict:
        n_local = n_for(name)
        big = n_local * STREAM_LEN * 2 >= (1 << 30)
        s_steps = args.sub_steps or (20 if big else 200)
        s_warm = 3 if big else 20
        shs = Shard(ctx, name, n_local)
        srec, saux = measure(ctx, shs, s_steps, s_warm, min(args.preroll_ms, 100.0), args.gather_every, args.min_region_ms)
        attach_traffic(srec, name, n_local, src_hash)
        if world == 1 and not args.no_cpu_baseline:
            ns = min(n_local, args.sub_cpu_sample)
            srec["match_rate"], _, _ = oracle_match(shs, saux["res"], saux["got_payloads"], shs.inputs[0], ns, cores)
            srec["match_sample_streams"] = ns
            if len(set(shs.bauds)) > 1:
                # a CPU figure per baud share: the reference is 2.8x slower per sample at 300 baud and
                # 1.4x faster at 2400 baud than at 1200 (the sync search costs (4096 - 2bf) * 2bf)
                by = {}
                for b_ in sorted(set(shs.bauds)):
                    idx = np.nonzero(shs.bf_h == 48000 // b_)[0]
                    doc, rate, nn = cpu_baseline_for(shs, saux["res"], saux["got_payloads"], min(ns, idx.size), cores,
                                                     3.0, idx=idx, label=f" (the first {b_}-baud streams)")
                    doc["match_rate"], doc["streams"] = rate, nn
                    by[str(b_)] = doc
                srec["cpu_baseline_by_baud"] = by
            if name == "config4":
                srec["ber_curve"] = ber_curve(ctx, shs, ns, cores)
                shs.regenerate(shs.snr_db)
        if world == 1 and "f2_gate" in next_rows and name == "config2":
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import bench_rows
            subs.setdefault("f2_gate", {})["captures_%d" % n_local] = bench_rows.measure_gate(ctx, shs, max(args.next_reps, 50))
        return srec

    for name in sub_names:
        srec = rider(name, lambda: config_sub(name))
        per_workload[name] = srec.get("value")
        subs[name] = srec
        checkpoint(f"sub-records after {name}")
    if world == 1 and next_rows:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        if "f3_wav_ingest" in next_rows:
            subs["f3_wav_ingest"] = rider("f3_wav_ingest", lambda: __import__("bench_rows").measure_wav_ingest(ctx, args.wav_files))
        if "f5_wav_egress" in next_rows:
            subs["f5_wav_egress"] = rider("f5_wav_egress", lambda: __import__("bench_rows").measure_wav_egress(ctx, args.wav_files))
        if "f2_chain" in next_rows:          # r6: gate -> burst_slots -> demod as one graph, on led-in captures
            subs["f2_chain"] = rider("f2_chain", lambda: __import__("bench_rows").measure_gate_chain(ctx, args.next_reps))
        if "ragged_lengths" in next_rows:    # r6, on request: a ragged batch against the same samples as 1 s streams
            subs["ragged_lengths"] = rider("ragged_lengths", lambda: __import__("bench_rows").measure_ragged(ctx, max(3, args.next_reps // 2)))
        checkpoint("the per-rate tables")
        # (the steady-state table first: it follows the host-bound file rows, during which the GPU idles, so its 36
        # workloads start from a state closer to the headline's than after the 36 prerolls of the 4096-stream table)
        for row, n_str in sorted(RATES_ROWS.items(), key=lambda kv: -kv[1]):
            if row in next_rows:
                big = n_str >= 32768
                subs[row] = rider(row, lambda: __import__("bench_rows").measure_rates(
                    ctx, args.rates_steps or (20 if big else 120), n_str, warmup=3 if big else 10))
                checkpoint(f"what follows {row}")
    if world == 1 and subs and not args.workload and args.sub is None:
        # The headline is necessarily the first allocation of the process, and a given 6.29 GB allocation streams at its
        # own rate (+-2 %: tools/order_probe.py).  Here the same workload is measured AGAIN, last, in a recycled buffer
        # like every sub-record -- reported next to the headline, never instead of it.
        def again():
            shl = Shard(ctx, main_name, n_for(main_name))
            lrec, _ = measure(ctx, shl, args.steps, args.warmup, 50.0, 0, args.min_region_ms)
            return {"value": lrec["value"], "ms_per_step": lrec["ms_per_step"],
                    "frac": lrec["roofline"]["frac"], "kernel_ms": lrec["roofline"]["kernel_ms"],
                    "roundtrip_match_rate": lrec["roundtrip_match_rate"],
                    "note": "the headline workload measured again after all sub-records, in a recycled buffer"}
        out["headline_again_at_end"] = rider("headline_again_at_end", again)
    if subs:
        out["sub_records"] = subs
    out["per_workload_value"] = per_workload
    out["scaling_note"] = ("the headline workload is the config5 shard (65536 streams x 1 s @1200 baud per GPU, what north_star "
                           "names) at EVERY N: value(N) / (N * value(1)) is the weak-scaling efficiency; value_no_gather is the "
                           "same K-step regions without the exchange of the decoded records, gather_ms the duration of one "
                           "collective; config2 / config3 / config4, the SURVEY 8(f) rows f1 / f2 / f3 and rates_65536 ride along "
                           "as sub_records at N = 1 (f5_wav_egress and rates_4096: --sub), the config2 shard at N > 1")
    hb.beat("closing the process group" if ctx.use_dist else "writing the result")
    if ctx.use_dist:
        dist.destroy_process_group()
    # RCCL writes its version banner through C stdio, which would otherwise be flushed at exit,
    # AFTER the result: flush it now so that the JSON line is the last thing on stdout
    try:
        C.CDLL(None).fflush(None)
    except Exception:  # noqa: BLE001
        pass
    dog.stand_down()
    if rank == 0:
        # only the default run (headline + every rider) writes bench_full_n<N>.json; side runs get their own file
        path = write_full_record(out, world, tag)
        print(json.dumps(compact_line(out, path)), flush=True)
    hb.beat("done")
    if rank == 0 and os.environ.get("AFSK_BENCH_LAUNCHER") != "1":
        import shutil
        shutil.rmtree(hb.dir, ignore_errors=True)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="", choices=[""] + sorted(WORKLOADS),
                    help="headline workload (default: config5 at every N); "
                         "giving one explicitly drops the sub-records unless --sub lists them")
    ap.add_argument("--sub", default=None, help="comma list of sub-records: workloads and/or f1_modulate,f2_gate,f3_wav_ingest,f5_wav_egress,rates_4096,rates_65536 ('' = none)")
    ap.add_argument("--min-region-ms", type=float, default=50.0,
                    help="repeat the K-step timed region until the regions add up to this (0 = exactly one region)")
    ap.add_argument("--entry", default="auto", choices=["auto", "mixed"],
                    help="auto = one-rate shards use afsk_demod_batch_uniform, several rates afsk_demod_batch_grouped; "
                         "mixed = always the per-stream entry afsk_demod_batch (A/B)")
    ap.add_argument("--next-reps", type=int, default=20, help="timed launches of the f1 / f2 sub-records")
    ap.add_argument("--wav-files", type=int, default=4096, help="files of the f3_wav_ingest sub-record")
    ap.add_argument("--sub-steps", type=int, default=0, help="timed steps of every sub-record (0 = 20 / 200)")
    ap.add_argument("--rates-steps", type=int, default=0, help="timed steps per rate of rates_4096 / rates_65536 (0 = 120 / 20)")
    ap.add_argument("--cpu-budget-s", type=float, default=10.0, help="wall-clock budget of the headline cpu_baseline leg")
    ap.add_argument("--sub-cpu-sample", type=int, default=1024,
                    help="streams per sub-record (and per SNR of the BER curve) decoded by the CPU oracle")
    ap.add_argument("--streams", type=int, default=0, help="override streams per GPU")
    ap.add_argument("--bauds", default="", help="with --workload custom: comma list of baud rates cycled over the streams")
    ap.add_argument("--training-time", type=float, default=0.5,
                    help="with --workload custom: the Transmitter's training_time in seconds (default 0.5, the reference's): "
                         "a shorter training sequence leaves more of the second to data symbols (more decoded bytes per stream)")
    ap.add_argument("--lead", default="", help="with --workload custom: lead-in before every frame -- N (samples, the same for every "
                    "stream) or 'random' (per stream 0 ... 2047, like config5_lead): low-level noise, arbitrary clock index")
    ap.add_argument("--rate-order", default="cycle", choices=["cycle", "blocks"],
                    help="custom workloads: baud of stream i = bauds[i %% k] (cycle) or contiguous blocks of n / k streams (blocks)")
    ap.add_argument("--preroll-ms", type=float, default=300.0,
                    help="untimed launches of the same kernel before the warm-up steps, so the "
                         "GPU clocks have settled (the first ~20 ms under load run 4-6 %% slower)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-streams", type=int, default=0)
    ap.add_argument("--gather-every", type=int, default=0,
                    help="N > 1: all-gather the decoded records of G steps with one collective; "
                         "0 = as many steps as take about 4 ms (at most 64)")
    ap.add_argument("--gather-mode", default="root", choices=["root", "all"],
                    help="root = decoded records are gathered to rank 0 (default); all = all_gather_into_tensor to every rank")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (the product); gloo = diagnostic, collectives staged through the host")
    ap.add_argument("--share-gpu0", action="store_true",
                    help="diagnostic: all ranks on device 0 (needs --dist-backend gloo), to exercise the N > 1 flow on a 1-GPU box")
    ap.add_argument("--force-gather", action="store_true",
                    help="exercise the RCCL gather path even at N=1 (single-rank group); diagnostics")
    ap.add_argument("--deadline-s", type=float, default=420.0,
                    help="wall-clock limit of the whole run (launcher: over its rank children; a rank: from its own start). "
                         "When it passes, ONE JSON line says what every rank was doing -- with the headline's numbers if it "
                         "had been measured -- and the exit code is non-zero.  0 = no limit")
    ap.add_argument("--pg-timeout-s", type=float, default=90.0,
                    help="torch.distributed process-group timeout: rendezvous and every collective")
    ap.add_argument("--no-value-no-gather", action="store_true",
                    help="N > 1: skip the extra K-step regions without the gather (value_no_gather)")
    args = ap.parse_args()

    # dmabuf IPC for RCCL on this pool (exported on the boxes already; set before anything initialises HIP)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    global RATE_ORDER, TRAINING_TIME
    RATE_ORDER = args.rate_order
    if args.bauds:
        bl = tuple(int(b) for b in args.bauds.split(","))
        WORKLOADS["custom"] = (WORKLOADS["custom"][0], bl, None, f"custom: streams x 1 s, clean, bauds {list(bl)}, per GPU")
    if args.training_time != 0.5:
        if args.workload != "custom":
            raise SystemExit("--training-time goes with --workload custom (the BASELINE configs use the reference's default)")
        TRAINING_TIME = args.training_time
        d = WORKLOADS["custom"]
        WORKLOADS["custom"] = (d[0], d[1], d[2], d[3] + f", training_time {args.training_time:g} s")
    if args.lead:
        if args.lead == "random":
            LEADS["custom"] = 2048
        else:
            LEAD_FIXED["custom"] = int(args.lead)
            if not 0 <= LEAD_FIXED["custom"] <= 4000:
                raise SystemExit("--lead: 0 ... 4000 samples (the frame must start inside the 4096-sample sync window)")
        d = WORKLOADS["custom"]
        WORKLOADS["custom"] = (d[0], d[1], d[2], d[3] + f", lead-in {args.lead}")
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        # bare `python bench.py --gpus N`: become the launcher.  This process never touches the GPU and never
        # imports torch: the devices are counted in sysfs (KFD topology).
        have = visible_gpus()
        if args.share_gpu0 and args.dist_backend != "gloo":
            sys.stderr.write("bench.py: --share-gpu0 needs --dist-backend gloo (RCCL refuses two ranks on one device)\n")
            raise SystemExit(2)
        if args.share_gpu0 and args.gpus > SHARE_GPU0_MAX_RANKS:
            sys.stderr.write(f"bench.py: --share-gpu0 runs at most {SHARE_GPU0_MAX_RANKS} ranks (the GPU boxes allow "
                             "6 processes per device, and the caller of this diagnostic usually holds the device too)\n")
            raise SystemExit(2)
        if have is not None and have < (1 if args.share_gpu0 else args.gpus):
            sys.stderr.write(f"bench.py: --gpus {args.gpus} but only {have} GPU(s) visible\n")
            raise SystemExit(2)
        raise SystemExit(self_launch(args.gpus, sys.argv[1:], deadline_s=args.deadline_s, steps=args.steps, warmup=args.warmup))
    run_rank(args)


if __name__ == "__main__":
    sys.modules.setdefault("bench", sys.modules["__main__"])   # tools/bench_rows.py imports this module by name
    main()
