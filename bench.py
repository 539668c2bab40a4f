#!/usr/bin/env python3
"""bench.py -- batched AFSK demodulation throughput on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path (afsk_demod_batch: sync search + symbol
correlator + squelch + Hamming decode + byte pack) over one batch of synthetic
streams that is already resident in HBM.  Default workload = BASELINE.json
configs[1]: 4096 Transmitter-generated, clean, 1 s, 1200-baud streams per GPU
(weak scaling: every rank demodulates its own 4096-stream shard; for N > 1 the
decoded records of each step are all-gathered over RCCL on a side stream,
overlapped with the next step's kernel).

Prints ONE JSON line on rank 0 (contract in the task statement) carrying
`roofline` (HBM-read bound; algorithmic bytes / HIP-event kernel time) and, at
N = 1, `cpu_baseline` (the CPU oracle -- a C port of the reference -- timed on
this box's host cores on a bounded sample, also used as the match-rate checker).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
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
}
STREAM_LEN = 48000
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="config2", choices=sorted(WORKLOADS))
    ap.add_argument("--streams", type=int, default=0, help="override streams per GPU")
    ap.add_argument("--preroll-ms", type=float, default=300.0,
                    help="untimed launches of the same kernel before the warm-up steps, so the "
                         "GPU clocks have settled (the first ~20 ms under load run 4-6 %% slower)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-streams", type=int, default=0)
    ap.add_argument("--gather-every", type=int, default=0,
                    help="N > 1: all-gather the decoded records of G steps with one collective; "
                         "0 = as many steps as take about 4 ms (at most 64)")
    ap.add_argument("--force-gather", action="store_true",
                    help="exercise the RCCL gather path even at N=1 (single-rank group); diagnostics")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from afskmodem_amd import _native, batch, synth
    from afskmodem_amd import dist as adist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    _native.require_device()          # no GPU -> loud failure, never a CPU fallback
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.force_gather
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    n_local, bauds, snr_db, desc = WORKLOADS[args.workload]
    if args.streams > 0:
        n_local = args.streams
    n_total = n_local * world
    first = rank * n_local          # this rank's contiguous shard (dist.shard_range of n_total)
    assert adist.shard_range(n_total, rank, world) == (first, first + n_local)

    # ---- synthesise this rank's shard on the device (not timed)
    gidx = np.arange(first, first + n_local)
    baud_arr = np.asarray([bauds[i % len(bauds)] for i in gidx], np.int32)
    bf_h = (48000 // baud_arr).astype(np.int32)
    plen_h = np.asarray([synth.ONE_SECOND_PAYLOAD[int(b)] for b in baud_arr], np.int32)
    pstride = int(plen_h.max())
    payload_h = synth.payload_bytes(2024, first, n_local, pstride)
    ts_h = np.asarray([synth.ts_cycles_for(int(b)) for b in baud_arr], np.int32)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    samples = torch.empty(n_local * STREAM_LEN, dtype=torch.int16, device=dev)
    off, ln = batch.uniform_layout(n_local, STREAM_LEN, dev)
    bf = t(bf_h)
    batch.modulate_batch(t(payload_h), t(plen_h), bf, t(ts_h), off, ln, STREAM_LEN, samples, True)
    if snr_db is not None:
        batch.add_noise_batch(samples, off, ln, STREAM_LEN, synth.snr_to_scale_q24(snr_db),
                              seed=99, stream_idx_base=first)
    torch.cuda.synchronize()

    stride = batch.out_stride_for(STREAM_LEN, int(bf_h.min()))
    # Output ring: 2 groups x G step slots, each slot one flat allocation (all six output arrays).
    # For N > 1 the decoded records of every step are exchanged, G steps per collective
    # ("fewer, larger collectives"): when a group of G steps has been launched, its G flat
    # buffers -- contiguous in memory -- are all-gathered on a side stream while the other
    # group is being filled.
    # Each collective costs the compute stream ~40 us (cross-stream events around it; measured
    # with --force-gather: +37 us per step at G = 1, +1 us per step at G = 64), so a group
    # should cover a few ms of kernels: 64 steps of config #2, 3-4 steps of the 6.29 GB configs.
    if args.gather_every > 0:
        G = args.gather_every
    else:
        est_step_s = 2.0 * n_local * STREAM_LEN / 6.0e12
        G = max(1, min(64, int(4e-3 / est_step_s)))
    _, flat_sz = batch.flat_layout(n_local, stride)
    group_flat = [torch.zeros(G * flat_sz, dtype=torch.uint8, device=dev) for _ in range(2)]
    outs = [[batch.views_of_flat(gf[k * flat_sz: (k + 1) * flat_sz], n_local, stride)
             for k in range(G)] for gf in group_flat]

    lib = _native.lib()
    cur = torch.cuda.current_stream()
    sptr = C.c_void_p(cur.cuda_stream)

    # argument tuples are built once per output slot: the timed loop is one ctypes call per step
    fn = lib.afsk_demod_batch
    slot_args = {id(o): (samples.data_ptr(), off.data_ptr(), ln.data_ptr(), bf.data_ptr(), 14000,
                         n_local, o.bytes.data_ptr(), stride, o.nbytes.data_ptr(), o.nbits.data_ptr(),
                         o.clock_idx.data_ptr(), o.term_frame.data_ptr(), o.status.data_ptr(), sptr)
                 for grp_outs in outs for o in grp_outs}

    def launch(o) -> None:
        rc = fn(*slot_args[id(o)])
        if rc != 0:
            _native.check(rc)

    comm = torch.cuda.Stream(device=dev) if use_dist else None
    gath_bufs = ([torch.empty(world * G * flat_sz, dtype=torch.uint8, device=dev)
                  for _ in range(2)] if use_dist else None)
    ready_ev = [torch.cuda.Event() for _ in range(2)]
    done_ev = [torch.cuda.Event() for _ in range(2)]
    gathers = 0
    last_slot = (0, 0)

    gather_timing = []        # (start, end) events of the collectives inside the timed region
    timing_on = False

    def gather_group(grp: int) -> None:
        nonlocal gathers
        ready_ev[grp].record(cur)
        comm.wait_event(ready_ev[grp])
        with torch.cuda.stream(comm):
            pair = None
            if timing_on and len(gather_timing) < 256:
                pair = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                pair[0].record(comm)
            dist.all_gather_into_tensor(gath_bufs[grp], group_flat[grp])
            if pair is not None:
                pair[1].record(comm)
                gather_timing.append(pair)
            done_ev[grp].record(comm)
        gathers += 1

    def step(i: int) -> None:
        """Launch the demod of step i into slot (i % G) of group (i // G) % 2; after the last
        slot of a group, gather the whole group on the comm stream (overlaps the next group)."""
        nonlocal last_slot
        grp, k = (i // G) & 1, i % G
        if comm is not None and k == 0 and i >= 2 * G:
            cur.wait_event(done_ev[grp])   # the gather that read this group 2G steps ago is done
        launch(outs[grp][k])
        last_slot = (grp, k)
        if comm is not None and k == G - 1:
            gather_group(grp)

    def finish(n_steps: int) -> None:
        """Gather a trailing partial group so that every step's records have been exchanged."""
        if comm is not None and n_steps % G != 0:
            gather_group(((n_steps - 1) // G) & 1)

    def fence() -> None:
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    # Pre-roll: the device needs ~20 ms of sustained load before its clocks settle (kernel trace:
    # 67 us per launch falling to 63.5 us over the first ~250 launches of config #2,
    # profiles/README.md).  Same kernel, same buffers, not timed; then the W warm-up steps.
    preroll_launches = 0
    if args.preroll_ms > 0:
        t_pre = time.perf_counter()
        for _ in range(4):
            launch(outs[0][0])
        torch.cuda.synchronize()
        est = max((time.perf_counter() - t_pre) / 4, 1e-5)
        preroll_launches = int(args.preroll_ms * 1e-3 / est) + 1
        for _ in range(preroll_launches):
            launch(outs[0][0])
        torch.cuda.synchronize()
    for i in range(args.warmup):
        step(i)
    finish(args.warmup)
    fence()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    timing_on = True
    t0 = time.perf_counter()
    ev0.record(cur)
    for i in range(args.steps):
        step(i)
    finish(args.steps)
    ev1.record(cur)
    host_issue_s = time.perf_counter() - t0        # host time to enqueue the whole timed region
    fence()
    elapsed = time.perf_counter() - t0
    kernel_ms = ev0.elapsed_time(ev1) / max(args.steps, 1)   # avg launch duration incl. gaps
    if use_dist:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # ---- check + algorithmic bytes from the last step's outputs
    res = outs[last_slot[0]][last_slot[1]].cpu()
    got_payloads = res.payloads()
    if snr_db is None:
        ok = sum(got_payloads[s] == payload_h[s, : plen_h[s]].tobytes() for s in range(n_local))
        roundtrip_rate = ok / n_local
    else:
        roundtrip_rate = None
    # samples the reference must read: up to and including the squelch-triggering symbol
    active = np.minimum(np.maximum(res.term_frame.astype(np.int64)
                                   + (res.nbits.astype(np.int64) + 1) * bf_h, 4096), STREAM_LEN)
    out_bytes_alg = int(np.minimum(res.nbytes, stride).sum()) + 20 * n_local
    alg_bytes = int(2 * active.sum()) + out_bytes_alg
    achieved_gbs = alg_bytes / (kernel_ms * 1e-3) / 1e9
    traffic = None
    tfile = os.path.join(ROOT, "profiles", "traffic_latest.json")
    if os.path.exists(tfile):
        try:
            tj = json.load(open(tfile))
            if tj.get("workload") == args.workload and tj.get("streams") == n_local:
                traffic = tj.get("hbm_bytes_per_launch")
        except Exception:  # noqa: BLE001
            traffic = None

    samples_per_step = n_total * STREAM_LEN
    value = samples_per_step * args.steps / elapsed / 1e6

    out = {
        "metric": "Msamples/s demodulated (batched 48 kHz streams) + decoded-byte match rate vs CPU ref",
        "value": round(value, 1),
        "unit": "Msamples/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "preroll_launches": preroll_launches,
        "ms_per_step": round(elapsed / max(args.steps, 1) * 1e3, 5),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "int16",
        "data": "synthetic",
        "config": {"workload": desc, "streams_per_gpu": n_local, "streams_total": n_total,
                   "stream_len": STREAM_LEN, "bauds": list(bauds), "snr_db": snr_db,
                   "parallelism": f"stream-sharded x{world}" + (" + RCCL all-gather of decoded records" if world > 1 else "")},
        "roofline": {"bound": "hbm", "achieved": round(achieved_gbs, 1), "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": round(achieved_gbs / HBM_PEAK_GBS, 4),
                     "traffic": traffic,
                     "algorithmic_bytes_per_launch": alg_bytes,
                     "kernel_ms": round(kernel_ms, 5),
                     "full_buffer_gbs": round((2 * n_local * STREAM_LEN) / (kernel_ms * 1e-3) / 1e9, 1)},
        "roundtrip_match_rate": roundtrip_rate,
        "host_issue_ms_per_step": round(host_issue_s / max(args.steps, 1) * 1e3, 5),
    }

    if world == 1 and not args.no_cpu_baseline:
        from oracle import afsk_oracle as O   # checker + reported CPU baseline only
        cores = os.cpu_count() or 1
        ns = args.cpu_sample_streams or min(n_local, 2048)
        h = samples[: ns * STREAM_LEN].cpu().numpy()
        h_off = np.arange(ns, dtype=np.int64) * STREAM_LEN
        h_ln = np.full(ns, STREAM_LEN, np.int32)
        t1 = time.perf_counter()
        want1 = O.demod_batch(h[: (ns // 8) * STREAM_LEN], h_off[: ns // 8], h_ln[: ns // 8],
                              bf_h[: ns // 8], 14000, out_stride=stride, n_threads=1)
        dt1 = time.perf_counter() - t1
        reps = 0
        t2 = time.perf_counter()
        while True:
            want = O.demod_batch(h, h_off, h_ln, bf_h[:ns], 14000, out_stride=stride, n_threads=cores)
            reps += 1
            if time.perf_counter() - t2 > 10.0 or reps >= 50:
                break
        dtc = (time.perf_counter() - t2) / reps
        match = 0
        for s in range(ns):
            nb = int(want["nbytes"][s])
            same = (nb == int(res.nbytes[s]) and int(want["nbits"][s]) == int(res.nbits[s])
                    and int(want["clock_idx"][s]) == int(res.clock_idx[s])
                    and want["bytes"][s, : min(nb, stride)].tobytes() == got_payloads[s][: min(nb, stride)])
            match += bool(same)
        del want1
        # "reference-shaped" figure: the pure-Python restatement (oracle/pyref.py) on one core;
        # in the dev container it runs at 1.01-1.07x the speed of the real afskmodem.py.
        from oracle import pyref
        npy = 3
        t3 = time.perf_counter()
        for s_i in range(npy):
            data, _, _, _ = pyref.demod(h[s_i * STREAM_LEN: (s_i + 1) * STREAM_LEN].tolist(), int(bf_h[s_i]))
            assert data == got_payloads[s_i], "pure-Python restatement disagrees with the GPU"
        dtp = (time.perf_counter() - t3) / npy
        out["cpu_baseline"] = {
            "value": round(ns * STREAM_LEN / dtc / 1e6, 1), "unit": "Msamples/s", "cores": cores,
            "kind": "port",
            "sample": f"first {ns} streams of the same batch, CPU oracle (C port of afskmodem.py hot path), "
                      f"{cores} threads, {reps} reps; single thread on {ns // 8} streams: "
                      f"{round((ns // 8) * STREAM_LEN / dt1 / 1e6, 1)} Msamples/s",
            "single_thread_value": round((ns // 8) * STREAM_LEN / dt1 / 1e6, 1),
            "python_reference_shaped_value": round(STREAM_LEN / dtp / 1e6, 3),
            "python_reference_shaped_note": "oracle/pyref.py (pure-Python restatement, CPython, 1 core, "
                                            f"{npy} streams); calibrated at 1.01-1.07x the real reference's "
                                            "speed in the dev container (DESIGN.md 4.3)",
        }
        out["match_rate"] = match / ns
        out["match_sample_streams"] = ns

    if use_dist:
        # the gathered copy of this rank's last step must equal its own outputs
        grp, k = last_slot
        mine = gath_bufs[grp][rank * G * flat_sz + k * flat_sz: rank * G * flat_sz + (k + 1) * flat_sz]
        out["gather_check"] = bool(torch.equal(mine, group_flat[grp][k * flat_sz: (k + 1) * flat_sz]))
        out["gathers_in_timed_region"] = (args.steps + G - 1) // G
        if gather_timing:
            # duration of the RCCL all-gather itself (comm stream, overlapped with the next
            # group's kernels), reported separately as SURVEY 8(d) config 5 asks
            gms = sorted(a.elapsed_time(b) for a, b in gather_timing)
            out["gather_ms"] = {"median": round(gms[len(gms) // 2], 4), "max": round(gms[-1], 4),
                                "bytes_per_rank": int(G * flat_sz), "measured": len(gms)}
        out["config"]["gather_every_steps"] = G
    if rank == 0:
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
