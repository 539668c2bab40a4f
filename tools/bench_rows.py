#!/usr/bin/env python3
"""bench_rows.py -- the rider rows of bench.py: the SURVEY 8(f) rows f1 (on-device modulator), f2 (live-gate
replay), f3 (.wav ingest), the egress mirror f5, and the per-rate tables rates_4096 / rates_65536.

They live outside bench.py so that the file the driver runs -- launcher, headline, configs 2-4, cpu_baseline --
stays small, and bench.py calls every rider inside try / except: a failing rider becomes an {"error": ...}
sub-record and can never cost the parsed headline line.  `python bench.py --sub f1_modulate,f2_gate,...` selects
them; the default N = 1 run carries f1, f2, f3 and rates_65536.

oracle/ is imported here as the checker only, after the timed launches."""
from __future__ import annotations

import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import bench  # noqa: E402
from bench import (ALL_RATES, HBM_PEAK_GBS, STREAM_LEN, Ctx, Shard, measure, median, oracle_match,  # noqa: E402
                   usable_cpus)

def event_timed(torch, stream, launch, reps: int, warm: int = 3):
    """reps launches on `stream`, one HIP-event interval per launch -> (avg_ms, median_ms, all)."""
    for _ in range(warm):
        launch()
    torch.cuda.synchronize()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    marks[0].record(stream)
    for k in range(reps):
        launch()
        marks[k + 1].record(stream)
    torch.cuda.synchronize()
    ms = [marks[k].elapsed_time(marks[k + 1]) for k in range(reps)]
    return sum(ms) / len(ms), median(ms), ms


def roofline_obj(alg_bytes: int, avg_ms: float, med_ms: float, bound: str = "hbm", peak: float = HBM_PEAK_GBS) -> dict:
    ach = alg_bytes / (avg_ms * 1e-3) / 1e9
    return {"bound": bound, "achieved": round(ach, 1), "peak": round(peak, 1), "unit": "GB/s",
            "frac": round(ach / peak, 4), "traffic": None, "algorithmic_bytes_per_launch": int(alg_bytes),
            "kernel_ms": round(avg_ms, 5), "kernel_ms_median": round(med_ms, 5),
            "frac_at_median": round(alg_bytes / (med_ms * 1e-3) / 1e9 / peak, 4)}


def measure_modulate(ctx: Ctx, sh: Shard, reps: int, check_streams: int = 64) -> dict:
    """the on-device modulator (Transmitter.__getFrames ref:452-469 + ECC.encode
    ref:166-175 + the .wav writer quirk ref:239-244) re-writing the shard's whole input buffer.
    Write-bound: 2 B per sample out (+ the payload bytes and 24 B of per-stream metadata in)."""
    from afskmodem_amd import _native
    torch = ctx.torch
    x = sh.inputs[0]
    sptr = C.c_void_p(ctx.cur.cuda_stream)
    quirk = 0 if 12000 in sh.bauds else 1
    args_ = (sh._payload_d.data_ptr(), int(sh._payload_d.shape[1]), sh._plen_d.data_ptr(), sh.bf.data_ptr(),
             sh._ts_d.data_ptr(), sh.off.data_ptr(), sh.ln.data_ptr(), STREAM_LEN, sh.n_local, quirk, x.data_ptr(), sptr)

    def launch():
        rc = ctx.lib.afsk_modulate_batch(*args_)
        if rc != 0:
            _native.check(rc)

    x.zero_()
    avg, med, _ = event_timed(torch, ctx.cur, launch, reps)
    alg = 2 * sh.n_local * STREAM_LEN + int(sh.plen_h.sum()) + 24 * sh.n_local
    rec = {"row": "f1 on-device modulator (afsk_modulate_batch)", "streams": sh.n_local, "stream_len": STREAM_LEN,
           "bauds": list(sh.bauds), "launches": reps, "unit": "Msamples/s",
           "value": round(sh.n_local * STREAM_LEN / (avg * 1e-3) / 1e6, 1),
           "roofline": roofline_obj(alg, avg, med)}
    rec["roofline"]["bound_note"] = "HBM WRITE bound: 2 B per sample stored once"
    if not ctx.args.no_cpu_baseline:
        from oracle import afsk_oracle as O   # checker only, after the timed launches
        ns = min(check_streams, sh.n_local)
        got = x[: ns * STREAM_LEN].cpu().numpy().reshape(ns, STREAM_LEN)
        want = O.modulate_batch(sh.payload_h[:ns], sh.plen_h[:ns], sh.bf_h[:ns],
                                np.asarray([int(t) for t in sh._ts_d[:ns].cpu().numpy()], np.int32),
                                np.arange(ns, dtype=np.int64) * STREAM_LEN, np.full(ns, STREAM_LEN, np.int32),
                                ns * STREAM_LEN, bool(quirk)).reshape(ns, STREAM_LEN)
        rec["oracle_match_rate"] = float((got == want).all(axis=1).mean())
        rec["oracle_sample_streams"] = ns
    return rec


def measure_gate(ctx: Ctx, sh: Shard, reps: int, check_streams: int = 64) -> dict:
    """live-gate replay (Receiver.__listen ref:299-319: 2048-frame block amplitudes,
    start > 18000, stop < 14000) over the shard's streams as captures.  Read-bound: 2 B per sample of
    every whole 2048-frame block in (+ 4 B per block and 12 B per capture out)."""
    from afskmodem_amd import _native
    torch = ctx.torch
    n = sh.n_local
    max_blocks, max_bursts = STREAM_LEN // 2048, 4
    i32 = lambda *shape: torch.zeros(shape, dtype=torch.int32, device=ctx.dev)  # noqa: E731
    amp, nb, bs, bl, oe = i32(n, max_blocks), i32(n), i32(n, max_bursts), i32(n, max_bursts), i32(n)
    sptr = C.c_void_p(ctx.cur.cuda_stream)
    nin = len(sh.inputs)
    calls = [(x.data_ptr(), sh.off.data_ptr(), sh.ln.data_ptr(), STREAM_LEN, 18000, 14000, n, max_bursts,
              amp.data_ptr(), nb.data_ptr(), bs.data_ptr(), bl.data_ptr(), oe.data_ptr(), sptr) for x in sh.inputs]
    k = [0]

    def launch():
        rc = ctx.lib.afsk_gate_batch(*calls[k[0] % nin])
        k[0] += 1
        if rc != 0:
            _native.check(rc)

    avg, med, _ = event_timed(torch, ctx.cur, launch, reps)
    alg = 2 * n * max_blocks * 2048 + 4 * n * max_blocks + 12 * n
    rec = {"row": "f2 live-gate replay (afsk_gate_batch: block amplitudes + burst scan)", "captures": n,
           "capture_len": STREAM_LEN, "launches": reps, "input_buffers_rotated": nin, "unit": "Msamples/s",
           "value": round(n * STREAM_LEN / (avg * 1e-3) / 1e6, 1),
           "bursts_found": int(nb.sum().item()),
           "roofline": roofline_obj(alg, avg, med)}
    if not ctx.args.no_cpu_baseline:
        from oracle import afsk_oracle as O   # checker only
        ns = min(check_streams, n)
        h = sh.inputs[(k[0] - 1) % nin][: ns * STREAM_LEN].cpu().numpy().reshape(ns, STREAM_LEN)
        g_nb, g_bs, g_bl, g_oe = (t[:ns].cpu().numpy() for t in (nb, bs, bl, oe))
        ok = 0
        for i in range(ns):
            bursts, open_end = O.gate_stream(h[i], 18000, 14000, max_bursts)
            k_ = int(g_nb[i])
            ok += bool(len(bursts) == k_ and open_end == int(g_oe[i])
                       and bursts == [(int(g_bs[i, j]), int(g_bl[i, j])) for j in range(k_)])
        rec["oracle_match_rate"] = ok / ns
        rec["oracle_sample_streams"] = ns
    return rec


def bench_tmpdir(prefix: str, need_bytes: int) -> str:
    """A scratch directory for the file rows (f3 / f5): on tmpfs (/dev/shm) when there is room -- the boxes' /tmp is an
    overlay on a disk, where NEW files run into the kernel's dirty-page throttling from the second GB on (measured:
    14 ms for the first 4096 x 96 KB files of a box, 350+ ms for every later batch) -- else wherever tempfile puts it."""
    import shutil
    import tempfile
    forced = os.environ.get("AFSK_BENCH_TMPDIR")        # (tests point this at a directory that cannot exist)
    if forced:
        return tempfile.mkdtemp(prefix=prefix, dir=forced)
    try:
        if os.path.isdir("/dev/shm") and shutil.disk_usage("/dev/shm").free > 4 * need_bytes + (1 << 30):
            return tempfile.mkdtemp(prefix=prefix, dir="/dev/shm")
    except OSError:
        pass
    return tempfile.mkdtemp(prefix=prefix)


def measure_wav_ingest(ctx: Ctx, n_files: int = 4096, reps: int = 5) -> dict:
    """n .wav files (SoundInput.loadFromFile ref:213-217) -> the stream-major
    device layout (afsk_file_sizes + afsk_wav_ingest: one open / header walk / pread / close per file,
    pipelined against the H2D copies), then decoded by Receiver.load_batch.
    PCIe-bound: measured against ONE pinned hipMemcpy of the same byte count on this box."""
    import shutil
    import tempfile
    import afskmodem_amd as afskmodem
    from afskmodem_amd import batch
    torch = ctx.torch
    afskmodem.LOG_LEVEL = 5
    d = bench_tmpdir("afsk_bench_wavs_", n_files * 96044)
    try:
        t = afskmodem.Transmitter(1200)
        payloads = [bytes([48 + i]) * 34 for i in range(16)]
        for i, pl in enumerate(payloads):
            t.save(pl, os.path.join(d, f"seed{i}.wav"))
        names = []
        for i in range(n_files):
            fn = os.path.join(d, f"f{i:05d}.wav")
            shutil.copyfile(os.path.join(d, f"seed{i % 16}.wav"), fn)
            names.append(fn)
        total_bytes = sum(os.path.getsize(f) for f in names)

        def timed(fn_):
            ts = []
            for _ in range(reps):
                t0 = time.perf_counter()
                fn_()
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
            return ts

        batch.load_wav_batch(names, ctx.dev)                       # warm, untimed: I/O pool, pinned buffers, torch's
        torch.cuda.synchronize()                                   # allocator (a first 393 MB block is a hipMalloc)
        ing = timed(lambda: batch.load_wav_batch(names, ctx.dev))
        pin = torch.empty(total_bytes // 2, dtype=torch.int16, pin_memory=True)
        devbuf = torch.empty(total_bytes // 2, dtype=torch.int16, device=ctx.dev)
        devbuf.copy_(pin, non_blocking=True)
        torch.cuda.synchronize()
        pc = timed(lambda: devbuf.copy_(pin, non_blocking=True))
        rx = afskmodem.Receiver(1200)
        e2e = timed(lambda: rx.load_batch(names, string=False))
        decoded = rx.load_batch(names, string=False)
        ok = sum(decoded[i] == payloads[i % 16] for i in range(n_files))
        avg, med = sum(ing) / len(ing), median(ing)
        peak = total_bytes / min(pc) / 1e9
        rec = {"row": "f3 .wav ingest (afsk_file_sizes + afsk_wav_ingest: one pass per file)", "files": n_files,
               "bytes": total_bytes, "reps": reps, "unit": "files/s", "value": round(n_files / med),
               "ingest_ms": {"median": round(med * 1e3, 3), "best": round(min(ing) * 1e3, 3)},
               "pinned_hipMemcpy_ms": round(min(pc) * 1e3, 3),
               "load_batch_end_to_end_ms": {"median": round(median(e2e) * 1e3, 3), "best": round(min(e2e) * 1e3, 3)},
               "decoded_match_rate": ok / n_files,
               # host-side wall times (Python + syscalls + H2D): the MEDIAN call is the figure, the mean rides along
               "roofline": {"bound": "pcie", "achieved": round(total_bytes / med / 1e9, 2), "peak": round(peak, 2),
                            "unit": "GB/s", "frac": round(total_bytes / med / 1e9 / peak, 4), "traffic": None,
                            "algorithmic_bytes_per_launch": total_bytes, "kernel_ms": round(med * 1e3, 3),
                            "kernel_ms_mean": round(avg * 1e3, 3),
                            "bound_note": "host -> device link: peak = one pinned hipMemcpy of the same bytes measured in this "
                                          "run (best of %d); the ingest also stats, opens, walks, preads and closes every file "
                                          "(page cache warm: the files were just written)" % reps},
               "host_cores": os.cpu_count(), "usable_cpus": usable_cpus(), "numa_binding": ctx.numa, "files_on": os.path.dirname(d)}
        del pin, devbuf
        return rec
    finally:
        shutil.rmtree(d, ignore_errors=True)


def measure_wav_egress(ctx: Ctx, n_files: int = 4096, reps: int = 5) -> dict:
    """The mirror of f3 (r4): n streams of 1 s in device memory -> n canonical .wav files (afsk_wav_egress: D2H
    through the pinned ring, one open / pwritev / close per file; what Transmitter.save_batch uses).
    PCIe-bound: measured against ONE pinned device-to-host hipMemcpy of the same bytes on this box."""
    import shutil
    import tempfile
    import wave
    from afskmodem_amd import batch
    torch = ctx.torch
    d = bench_tmpdir("afsk_bench_out_", 2 * n_files * 96044)
    try:
        x = torch.randint(-32768, 32767, (n_files * STREAM_LEN,), dtype=torch.int16, device=ctx.dev)
        offs = np.arange(n_files, dtype=np.int64) * STREAM_LEN
        lens = np.full(n_files, STREAM_LEN, np.int32)
        # a corpus spread over 64 directories, and -- `one_directory` -- flat (on a disk-backed file system the two
        # differ a lot, on tmpfs hardly)
        for k in range(64):
            os.mkdir(os.path.join(d, f"d{k:02d}"))
        names = [os.path.join(d, f"d{i % 64:02d}", f"o{i:05d}.wav") for i in range(n_files)]
        flat_names = [os.path.join(d, f"o{i:05d}.wav") for i in range(n_files)]
        torch.cuda.synchronize()
        assert (batch.save_wav_batch(x, offs, lens, names) == 0).all()             # warm: creates the files
        ts, ts_over, ts_flat = [], [], []
        for _ in range(reps):                                  # NEW files: what a pipeline that produces a corpus does
            for fn in names:
                os.unlink(fn)
            t0 = time.perf_counter()
            st = batch.save_wav_batch(x, offs, lens, names)
            ts.append(time.perf_counter() - t0)
        for _ in range(max(2, reps // 2)):                     # NEW files, all in one directory
            t0 = time.perf_counter()
            batch.save_wav_batch(x, offs, lens, flat_names)
            ts_flat.append(time.perf_counter() - t0)
            for fn in flat_names:
                os.unlink(fn)
        for _ in range(reps):                                  # existing files of the same size, overwritten in place
            t0 = time.perf_counter()
            batch.save_wav_batch(x, offs, lens, names)
            ts_over.append(time.perf_counter() - t0)
        total_bytes = n_files * STREAM_LEN * 2
        pin = torch.empty(total_bytes // 2, dtype=torch.int16, pin_memory=True)
        pin.copy_(x, non_blocking=True)
        torch.cuda.synchronize()
        pc = []
        for _ in range(reps):
            t0 = time.perf_counter()
            pin.copy_(x, non_blocking=True)
            torch.cuda.synchronize()
            pc.append(time.perf_counter() - t0)
        ok = 0
        pick = list(range(0, n_files, max(1, n_files // 64)))
        for i in pick:
            with wave.open(names[i], "rb") as f:
                raw = f.readframes(f.getnframes())
            ok += raw == x[i * STREAM_LEN: (i + 1) * STREAM_LEN].cpu().numpy().tobytes()
        med, peak = median(ts), total_bytes / min(pc) / 1e9
        return {"row": "f5 .wav egress (afsk_wav_egress: device streams -> files)", "files": n_files, "bytes": total_bytes,
                "reps": reps, "unit": "files/s", "value": round(n_files / med), "all_status_ok": bool((st == 0).all()),
                "egress_ms": {"median": round(med * 1e3, 3), "best": round(min(ts) * 1e3, 3)},
                "egress_overwrite_in_place_ms": {"median": round(median(ts_over) * 1e3, 3), "best": round(min(ts_over) * 1e3, 3)},
                "egress_new_files_one_directory_ms": {"median": round(median(ts_flat) * 1e3, 3), "best": round(min(ts_flat) * 1e3, 3)},
                "pinned_hipMemcpy_d2h_ms": round(min(pc) * 1e3, 3), "decoded_match_rate": ok / len(pick),
                "roofline": {"bound": "pcie", "achieved": round(total_bytes / med / 1e9, 2), "peak": round(peak, 2), "unit": "GB/s",
                             "frac": round(total_bytes / med / 1e9 / peak, 4), "traffic": None,
                             "algorithmic_bytes_per_launch": total_bytes, "kernel_ms": round(med * 1e3, 3),
                             "frac_overwrite_in_place": round(total_bytes / median(ts_over) / 1e9 / peak, 4),
                             "bound_note": "NEW files spread over 64 directories (frac) / existing files overwritten in place (frac_overwrite_in_place); "
                                           "device -> host link: peak = one pinned hipMemcpy of the same bytes measured in this run; "
                                           "for new files the kernel's page allocation (24 pages per file), not the link, is the limit"},
                "files_on": os.path.dirname(d)}
    finally:
        shutil.rmtree(d, ignore_errors=True)


def measure_rates(ctx: Ctx, steps: int = 120, n_streams: int = 4096, check_streams: int = 64, warmup: int = 10) -> dict:
    """4096 x 1 s clean streams at EVERY rate a Receiver can be built for (bit_frames must divide 48000 and
    be a multiple of 4: 36 values, 12000 ... 24 baud), each through its own uniform kernel: time per launch,
    fraction of the HBM peak in algorithmic bytes, round trip to the modulated payloads and the CPU oracle
    on a sample.  (Below ~100 baud a 1 s stream holds 0 - 2 payload bytes: training, terminator and tail.)"""
    rows = {}
    cores = usable_cpus()
    for baud in ALL_RATES:
        sh = Shard(ctx, "custom", n_streams, bauds=(baud,), desc=f"{n_streams} streams x 1 s @{baud} baud, clean")
        # 50 ms pre-roll: every rate starts from settled clocks; the figure is the MEDIAN of three K-step regions
        # (single regions of one rate differ by +-0.03 from run to run)
        rec, aux = measure(ctx, sh, steps, warmup, 50.0, 0, 3.0 * steps * (0.9 if n_streams >= 32768 else 0.06), 3)
        row = {"bit_frames": 48000 // baud, "payload_bytes": int(sh.plen_h[0]), "entry": rec["entry"],
               "ms_per_step": rec["ms_per_step"], "kernel_ms": rec["roofline"]["kernel_ms"],
               "kernel_ms_median": rec["roofline"]["kernel_ms_median"], "frac": rec["roofline"]["frac"],
               "frac_at_median": rec["roofline"]["frac_at_median"],
               "algorithmic_bytes_per_launch": rec["roofline"]["algorithmic_bytes_per_launch"],
               "full_buffer_gbs": rec["roofline"]["full_buffer_gbs"],
               "roundtrip_match_rate": rec["roundtrip_match_rate"],
               "all_timed_steps_identical": rec["all_timed_steps_identical"]}
        if not ctx.args.no_cpu_baseline:
            row["match_rate"], _, _ = oracle_match(sh, aux["res"], aux["got_payloads"], sh.inputs[0],
                                                   min(check_streams, n_streams), cores)
        rows[str(baud)] = row
        # (no empty_cache(): the next rate's 6.29 GB buffer is the block this one gives back to torch's caching
        # allocator.  A FRESHLY hipMalloc'ed buffer streams 2-4 % slower for its first seconds -- tools/order_probe.py:
        # 0.824 right after allocation, 0.841 for the same buffer ten seconds later -- and that is not what
        # "inputs resident in HBM" means.)
        del sh, aux
    fr = [r["frac"] for r in rows.values()]
    slow = sorted(rows, key=lambda b: rows[b]["frac"])[:3]
    doc = {"row": f"{n_streams} x 1 s clean streams at each of the {len(rows)} rates a Receiver can be built for "
                  "(afsk_demod_batch_uniform: one kernel per bit_frames)",
           "steps": steps, "min_frac": min(fr), "max_frac": max(fr), "median_frac": median(fr),
           "rates_below_0.60": [b for b, r in rows.items() if r["frac"] < 0.60],
           "rates_below_0.75": [b for b, r in rows.items() if r["frac"] < 0.75],
           "slowest": {b: rows[b]["frac"] for b in slow},
           "all_round_trips_exact": all(r["roundtrip_match_rate"] == 1.0 for r in rows.values()),
           "by_baud": rows}
    if not ctx.args.no_cpu_baseline:
        doc["min_match_rate"] = min(r["match_rate"] for r in rows.values())
    return doc


def measure_gate_chain(ctx: Ctx, reps: int = 20, n_captures: int = 65536, max_bursts: int = 1, check_streams: int = 1024) -> dict:
    """f2 -> demod END TO END (r6): 65536 one-second captures whose burst starts anywhere inside the first 2048-frame
    block (the config5_lead shard: lead-in noise, then a Transmitter frame) -> afsk_gate_batch (Receiver.__listen,
    ref:299-319) -> GateResult.burst_slots (fixed slots, no host round trip) -> afsk_demod_batch_uniform on the bursts
    (what Receiver.receive would hand to __decodeBits, ref:402-417) -> decoded bytes, as ONE captured HIP graph.
    A burst the gate cuts at a block boundary starts in the middle of the training sequence: its clock index is
    arbitrary.  Algorithmic bytes: the gate reads every whole block of every capture, the demodulator the burst up to
    its squelch symbol (both algorithms need the samples: counted once each)."""
    from afskmodem_amd import batch
    torch = ctx.torch
    sh = Shard(ctx, "config5_lead", n_captures)
    n = sh.n_local
    x = sh.inputs[0]
    stride = batch.out_stride_for(STREAM_LEN, 40)
    out = batch.alloc_result(n * max_bursts, stride, ctx.dev)
    keep = {}

    def chain():
        g = batch.gate_batch(x, sh.off, sh.ln, STREAM_LEN, 18000, 14000, max_bursts, stream=ctx.cur)
        s_off, s_len = g.burst_slots(sh.off)
        batch.demod_batch(x, s_off, s_len, 40, 14000, out=out, stream=ctx.cur)
        keep["g"], keep["slots"] = g, (s_off, s_len)
        return g

    with torch.cuda.stream(ctx.cur):
        chain()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream(device=ctx.dev)
    side.wait_stream(ctx.cur)
    saved_cur = ctx.cur
    try:
        ctx.cur = side
        with torch.cuda.stream(side):
            with torch.cuda.graph(graph, stream=side):
                chain()
    finally:
        ctx.cur = saved_cur
    torch.cuda.synchronize()
    out.flat.zero_()
    with torch.cuda.stream(ctx.cur):
        avg, med, _ = event_timed(torch, ctx.cur, graph.replay, reps)
    # the parts, eagerly, for the split
    g = keep["g"]
    s_off, s_len = keep["slots"]
    with torch.cuda.stream(ctx.cur):
        g_avg, _, _ = event_timed(torch, ctx.cur, lambda: batch.gate_batch(x, sh.off, sh.ln, STREAM_LEN, 18000, 14000, max_bursts, stream=ctx.cur), reps)
        d_avg, _, _ = event_timed(torch, ctx.cur, lambda: batch.demod_batch(x, s_off, s_len, 40, 14000, out=out, stream=ctx.cur), reps)
    res = out.cpu()
    pays = res.payloads()
    nb = g.n_bursts.cpu().numpy()
    ok = sum(pays[s * max_bursts] == sh.payload_h[s, : sh.plen_h[s]].tobytes() for s in range(n))
    lens = s_len.cpu().numpy().astype(np.int64)
    active = np.where(res.status == 0, np.minimum(res.term_frame.astype(np.int64) + (res.nbits.astype(np.int64) + 1) * 40, lens), np.minimum(lens, 4096))
    blocks = STREAM_LEN // 2048
    alg_gate = 2 * n * blocks * 2048 + 4 * n * blocks + 12 * n
    alg_demod = int(2 * active.sum()) + int(np.minimum(res.nbytes, stride).sum()) + 20 * n * max_bursts
    rec = {"row": "f2 -> demod chain as ONE HIP graph: afsk_gate_batch -> burst_slots -> afsk_demod_batch_uniform (config5_lead captures)",
           "captures": n, "max_bursts": max_bursts, "launches": reps, "unit": "Msamples/s",
           "value": round(n * STREAM_LEN / (avg * 1e-3) / 1e6, 1),
           "chain_ms": round(avg, 5), "chain_ms_median": round(med, 5), "gate_ms_alone": round(g_avg, 5), "demod_ms_alone": round(d_avg, 5),
           "bursts_found": int(nb.sum()), "roundtrip_match_rate": ok / n,
           "clock_index_unaligned_share": round(float(((res.clock_idx[res.status == 0].astype(np.int64) * 2) & 15).astype(bool).mean()), 4),
           "roofline": roofline_obj(alg_gate + alg_demod, avg, med),
           "demod_frac_alone": round(alg_demod / (d_avg * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
           "gate_frac_alone": round(alg_gate / (g_avg * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
    if not ctx.args.no_cpu_baseline:
        from oracle import afsk_oracle as O   # checker only
        ns = min(check_streams, n)
        so, sl = s_off.cpu().numpy()[: ns * max_bursts], s_len.cpu().numpy()[: ns * max_bursts]
        hx = x[: ns * STREAM_LEN].cpu().numpy()
        want = O.demod_batch(hx, so, sl, np.full(ns * max_bursts, 40, np.int32), 14000, out_stride=stride, n_threads=usable_cpus())
        same = all(np.array_equal(getattr(res, f)[: ns * max_bursts], want[f]) for f in ("nbytes", "nbits", "clock_idx", "term_frame", "status"))
        rec["oracle_match_rate"] = 1.0 if same else 0.0
        rec["oracle_sample_streams"] = ns
    del graph
    return rec


def measure_ragged(ctx: Ctx, reps: int = 10, seconds: int = 65536, baud: int = 1200) -> dict:
    """Ragged batches (r6): stream lengths log-uniform in 0.25 ... 4 s, `seconds` seconds of audio in all, one rate --
    through (a) the plain uniform launch in stream order, (b) a length-aware plan (afsk_group_plan_create_ragged: the
    uniform kernel walks the streams longest first inside windows of 4096) -- against (c) the same number of samples as
    1 s streams of the same frame shape (training 0.1 s).  One wavefront decodes one stream whatever its length and a
    workgroup of four keeps its share of a CU until its longest stream ends."""
    from afskmodem_amd import batch, synth
    torch = ctx.torch
    bf = 48000 // baud
    rng = np.random.default_rng(4242)
    lens = []
    total = 0
    while total < seconds * STREAM_LEN:
        l = int(np.exp(rng.uniform(np.log(12000), np.log(192000)))) & ~7
        lens.append(l)
        total += l
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(ctx.dev)  # noqa: E731
    ts_c = synth.ts_cycles_for(baud, 0.1)

    def make(ln):
        n = ln.size
        plen = np.maximum(0, (ln.astype(np.int64) - ts_c * 2 * bf - 4 * bf - 4800) // (14 * bf)).astype(np.int32)
        payload = synth.payload_bytes(77, 0, n, max(1, int(plen.max())))
        off = np.concatenate([[0], np.cumsum(ln[:-1], dtype=np.int64)]).astype(np.int64)
        x = torch.empty(int(ln.astype(np.int64).sum()), dtype=torch.int16, device=ctx.dev)
        d_off, d_ln = t(off), t(ln)
        batch.modulate_batch(t(payload), t(plen), t(np.full(n, bf, np.int32)), t(np.full(n, ts_c, np.int32)), d_off, d_ln,
                             int(ln.max()), x, baud != 12000)
        stride = batch.out_stride_for(int(ln.max()), bf)
        return dict(x=x, off=d_off, ln=d_ln, n=n, plen=plen, payload=payload, stride=stride, out=batch.alloc_result(n, stride, ctx.dev))

    def run(b, plan, tag):
        def launch():
            batch.demod_batch(b["x"], b["off"], b["ln"], None if plan is not None else bf, 14000, out=b["out"],
                              plan=plan, entry="auto" if plan is not None else "uniform", stream=ctx.cur)
        b["out"].flat.zero_()
        with torch.cuda.stream(ctx.cur):
            # (a fresh 6 GB allocation and cold clocks: ~50 ms of the same launches first, like bench.measure's pre-roll)
            avg, med, _ = event_timed(torch, ctx.cur, launch, reps, warm=50)
        res = b["out"].cpu()
        pays = res.payloads()
        ok = sum(pays[s] == b["payload"][s, : b["plen"][s]].tobytes() for s in range(b["n"])) / b["n"]
        hl = b["ln"].cpu().numpy().astype(np.int64)
        active = np.minimum(np.maximum(res.term_frame.astype(np.int64) + (res.nbits.astype(np.int64) + 1) * bf, 4096), hl)
        alg = int(2 * active.sum()) + int(np.minimum(res.nbytes, b["stride"]).sum()) + 20 * b["n"]
        return {"what": tag, "streams": b["n"], "kernel_ms": round(avg, 5), "kernel_ms_median": round(med, 5),
                "frac": round(alg / (avg * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_launch": alg,
                "roundtrip_match_rate": ok}

    rag = make(np.asarray(lens, np.int32))
    plan = batch.GroupPlan(np.full(rag["n"], bf, np.int32), ctx.dev, stream_len=np.asarray(lens, np.int32))
    a = run(rag, None, "ragged, stream order (plain uniform launch)")
    b_ = run(rag, plan, "ragged, length-aware plan (longest first inside windows of 4096 streams)")
    a2 = run(rag, None, "ragged, stream order again")
    del rag, plan
    uni = make(np.full(total // STREAM_LEN, STREAM_LEN, np.int32))
    c = run(uni, None, "the same samples as 1 s streams")
    del uni
    per_byte = lambda r: r["kernel_ms"] / r["algorithmic_bytes_per_launch"]  # noqa: E731
    return {"row": f"ragged one-rate batch ({baud} baud, lengths log-uniform 0.25 ... 4 s, {total / 48000:.0f} s of audio) vs the same samples as 1 s streams",
            "launches": reps, "stream_order": a, "length_aware_plan": b_, "stream_order_again": a2, "uniform_1s": c,
            "ragged_over_uniform_time_per_byte_stream_order": round(per_byte(a) / per_byte(c), 4),
            "ragged_over_uniform_time_per_byte_planned": round(per_byte(b_) / per_byte(c), 4),
            "value": round(total / (b_["kernel_ms"] * 1e-3) / 1e6, 1), "unit": "Msamples/s",
            "roofline": {"frac": b_["frac"]}, "roundtrip_match_rate": min(a["roundtrip_match_rate"], b_["roundtrip_match_rate"], c["roundtrip_match_rate"])}
