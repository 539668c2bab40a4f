"""Multi-GPU path on real hardware (-m gpu; skipped on a 1-GPU box): two ranks over RCCL, each
demodulates its own shard of one seeded batch with the HIP kernel, `dist.gather_flat` exchanges
the decoded records, and EVERY rank checks EVERY rank's slice against the CPU oracle.
Streams are independent (reference afskmodem.py:354-381), so shard + gather must equal the
oracle's single-process decode of the whole batch, bit for bit."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FIELDS = ("nbytes", "nbits", "clock_idx", "term_frame", "status")
L = 48000


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _batch_meta(first, n):
    from afskmodem_amd import synth
    bauds = np.asarray([(300, 1200, 2400, 600)[i % 4] for i in range(first, first + n)], np.int32)
    bf = (48000 // bauds).astype(np.int32)
    plen = np.asarray([synth.ONE_SECOND_PAYLOAD[int(b)] for b in bauds], np.int32)
    payload = synth.payload_bytes(77, first, n, 68)
    ts = np.asarray([synth.ts_cycles_for(int(b)) for b in bauds], np.int32)
    snr = np.where((np.arange(first, first + n) % 5) == 0, 6.0, 40.0)
    return bf, plen, payload, ts, snr


def _rank_main(rank, world, port, n_total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    from afskmodem_amd import _native, batch, synth
    from afskmodem_amd import dist as adist
    from oracle import afsk_oracle as O   # checker only
    try:
        _native.require_device()
        torch.cuda.set_device(rank)
        dev = torch.device("cuda", rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        ones = torch.ones(1, dtype=torch.int32, device=dev)
        dist.all_reduce(ones)
        b, e = adist.shard_range(n_total, rank, world)
        n = e - b
        bf, plen, payload, ts, snr = _batch_meta(b, n)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
        x = torch.empty(n * L, dtype=torch.int16, device=dev)
        off, ln = batch.uniform_layout(n, L, dev)
        d_bf = t(bf)
        batch.modulate_batch(t(payload), t(plen), d_bf, t(ts), off, ln, L, x, True)
        q24 = np.asarray([synth.snr_to_scale_q24(s) for s in snr], np.int32)
        batch.add_noise_batch(x, off, ln, L, q24, seed=5, stream_idx_base=b)
        stride = batch.out_stride_for(L, 20)
        res = batch.alloc_result(n, stride, dev)
        batch.demod_batch(x, off, ln, d_bf, 14000, out=res)
        equal = n_total % world == 0
        # the general (ragged-shard) form: shards may differ by one stream -> padded rows, one collective
        full = adist.gather_results(res, n_total)
        parts = adist.gather_flat(res, n_total) if equal else None
        torch.cuda.synchronize()
        # the oracle decodes this rank's own inputs; the gathered slice of every OTHER rank is
        # checked by that rank's oracle result, exchanged as a second (checker-side) gather
        want = O.demod_batch(x.cpu().numpy(), np.arange(n, dtype=np.int64) * L, np.full(n, L, np.int32),
                             bf, 14000, out_stride=stride)
        wflat = batch.alloc_result(n, stride, dev)
        for f in FIELDS:
            getattr(wflat, f).copy_(t(want[f]))
        wflat.bytes.copy_(t(want["bytes"][:, :stride]))
        wfull = adist.gather_results(wflat, n_total)
        col = torch.arange(stride, device=dev)[None, :]

        def same(g, w):
            r_ok = all(bool(torch.equal(getattr(g, f), getattr(w, f))) for f in FIELDS)
            m = col < torch.clamp(w.nbytes, max=stride)[:, None]
            return r_ok and bool(((g.bytes == w.bytes) | ~m).all().item())

        ok = int(full.nbytes.shape[0]) == n_total and same(full, wfull)
        if equal:
            rparts = adist.gather_flat_to_root(res, n_total, dst=world - 1)           # records to ONE rank
            torch.cuda.synchronize()
            if rank == world - 1:
                ok = ok and len(rparts) == world and all(
                    bool(torch.equal(rparts[r].flat, parts[r].flat)) for r in range(world))
            else:
                ok = ok and rparts is None
            wparts = adist.gather_flat(wflat, n_total)
            ok = ok and len(parts) == world
            for r in range(world):
                ok = ok and same(parts[r], wparts[r])
                rb, re_ = adist.shard_range(n_total, r, world)
                ok = ok and all(bool(torch.equal(getattr(parts[r], f), getattr(full, f)[rb:re_])) for f in FIELDS)
        # clean streams of my shard decode to their payloads (not vacuous: the oracle agrees AND the data is right)
        mine = batch.DemodResult(*(getattr(full, f)[b:e] for f in ("bytes",) + FIELDS)).cpu().payloads()
        clean_ok = all(mine[s] == payload[s, : plen[s]].tobytes() for s in range(n) if snr[s] > 30)
        q.put((rank, bool(ok), bool(clean_ok), int(ones.item()), ""))
        dist.destroy_process_group()
    except Exception as exc:  # noqa: BLE001
        q.put((rank, False, False, 0, repr(exc)))
        raise


@pytest.mark.parametrize("n_total", [96, 97])
def test_two_rank_rccl_gather_matches_oracle(n_total):
    import torch
    import torch.multiprocessing as mp
    from afskmodem_amd import _native
    assert _native.device_count() > 0, "no HIP device: GPU tests need an MI355X"
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs (the driver's multi-GPU box); 1-GPU boxes run "
                    "test_single_rank_rccl_gather_roundtrip instead")
    world = 2                     # 97 streams: shards of 48 and 49 (dist.gather_results pads the rows)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_main, args=(r, world, port, n_total, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert sorted(g[0] for g in got) == list(range(world))
    for rank, ok, clean_ok, seen, err in got:
        assert err == "", err
        assert seen == world, f"rank {rank}: all_reduce saw {seen} ranks"
        assert ok, f"rank {rank}: a gathered slice differs from the oracle"
        assert clean_ok, f"rank {rank}: clean streams did not decode to their payloads"


@pytest.mark.parametrize("n_total", [48, 49])
def test_single_rank_rccl_gather_roundtrip(n_total):
    """1-GPU boxes: the same code path with a single-rank RCCL group (spawned, so the process
    group never leaks into the pytest process): gather_flat AND the general gather_results form
    (pack -> all_gather_into_tensor over RCCL -> trim -> unpack) on device tensors."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rank_main, args=(0, 1, _free_port(), n_total, q))
    p.start()
    rank, ok, clean_ok, seen, err = q.get(timeout=600)
    p.join(timeout=120)
    assert p.exitcode == 0 and err == "", err
    assert seen == 1 and ok and clean_ok


def test_host_entries_follow_the_current_device():
    """the host-entry scratch cache keeps pinned windows + events between
    calls; when the current device changes, the events must be recreated on the new device.
    Needs two GPUs."""
    import torch
    from afskmodem_amd import batch
    from oracle import afsk_oracle as O
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs")
    rng = np.random.default_rng(4)
    streams = [O.wav_convert(O.get_frames(rng.integers(0, 256, 20, dtype=np.uint8).tobytes(), 1200)) for _ in range(40)]
    want = None
    for dev in (0, 1, 0, 1):
        with torch.cuda.device(dev):
            got = batch.demod_host_arrays(streams, 40)      # gather entry: pinned windows + events
            one = batch.demod_host_flat(streams[0], [0], [len(streams[0])], 40)
        if want is None:
            want = got
        assert got.payloads() == want.payloads() and one.payloads()[0] == want.payloads()[0], dev


def test_bench_two_rank_flow_on_one_gpu():
    """`python bench.py --gpus 2` end to end from a bare interpreter on a 1-GPU box: the parent
    launches two ranks (torch.distributed.run) without touching the GPU, both ranks demodulate
    their shard on device 0, exchange the decoded records (diagnostic gloo backend: RCCL refuses two
    ranks on one device) and every rank verifies every rank's gathered slice; ONE JSON line, rc 0."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dist-backend", "gloo",
                        "--share-gpu0", "--workload", "config2", "--streams", "768", "--sub", "", "--steps", "9",
                        "--warmup", "2", "--preroll-ms", "0", "--gather-every", "4"],
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and len(lines[0]) < 4096, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2
    assert d["gather_check"] == [True, True] and d["gather_check_on_every_rank"] is True
    assert d["roundtrip_match_rate"] == 1.0 and d["all_timed_steps_identical"] is True
    assert d["config"]["streams_total"] == 1536 and d["gathers_in_timed_region"] == 3
    # the same K-step regions without the exchange, next to the headline (gather cost separately)
    assert d["value_no_gather"] > 0 and d["gather_ms"]["median"] > 0 and d["host_threads_per_rank"] >= 1
    assert "pg up" in p.stderr and "first collective done: 2 ranks answered" in p.stderr and len(p.stderr) < 4096


@pytest.mark.parametrize("fault,rc_name,needle", [("1:die:shard built", "EXIT_RANK_FAILED", "rank 1 -> exit code 7"),
                                                  ("1:hang:timing without", "EXIT_DEADLINE", "deadline of")])
def test_bench_failure_paths_end_in_one_diagnostic_line(fault, rc_name, needle):
    """The un-losable line, rehearsed on the real rank code: rank 1 dies after building its shard / hangs before its
    first timed region (fault injection), rank 0 waits in a collective that can never complete.  The launcher
    terminates both, prints ONE line (value null, which rank, what every rank was doing) and exits non-zero."""
    import json
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    t0 = time.monotonic()
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dist-backend", "gloo",
                        "--share-gpu0", "--workload", "config2", "--streams", "768", "--sub", "", "--steps", "9",
                        "--warmup", "2", "--preroll-ms", "0", "--deadline-s", "75"],
                       capture_output=True, text=True, timeout=600, env=dict(os.environ, AFSK_BENCH_FAULT=fault))
    took = time.monotonic() - t0
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and len(lines[0]) < 4096, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert p.returncode == getattr(bench, rc_name) and took < 200, (p.returncode, took, p.stderr[-1500:])
    assert d["value"] is None and d["n_gpus"] == 2 and needle in d["error"], d
    assert set(d["heartbeats"]) == {"0", "1"} and d["heartbeats"]["1"]["phase"].startswith(fault.split(":")[2])
    assert d["metric"] == bench.METRIC and d["printed_by"] == "launcher"


def _torchrun_bench(extra_env=None, extra_args=()):
    """bench.py launched the way the DRIVER launches it for N > 1: `python -m torch.distributed.run --nnodes=1
    --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P bench.py --gpus 2 ...` (no bench.py launcher process:
    the ranks' own watchdogs are all there is).  Diagnostic gloo backend, both ranks on device 0."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--dist-backend", "gloo",
           "--share-gpu0", "--workload", "config2", "--streams", "768", "--sub", "", "--steps", "9", "--warmup", "2",
           "--preroll-ms", "0", "--gather-every", "4", *extra_args]
    env = dict(os.environ, **(extra_env or {}))
    env.pop("AFSK_BENCH_LAUNCHER", None)
    return subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)


def test_bench_under_an_external_torchrun_like_the_driver():
    """The scaling run's launch shape, rehearsed: ONE JSON line (the last thing on stdout), rc 0, both ranks seen,
    value_no_gather next to value."""
    import json
    p = _torchrun_bench()
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1 and p.stdout.strip().splitlines()[-1] == lines[0], p.stdout[-1500:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["gather_check"] == [True, True]
    assert d["value_no_gather"] > 0 and d["gather_ms"]["median"] > 0 and d["roundtrip_match_rate"] == 1.0
    assert "incomplete" not in d and "error" not in d


def test_bench_under_an_external_torchrun_a_dying_rank_still_yields_the_line():
    """Rank 1 dies after building its shard (fault injection).  torchrun terminates rank 0 with SIGTERM while its main
    thread sits in a collective that can never complete: the watchdog THREAD of rank 0 prints the diagnostic line
    (value null, rank 1's last phase, its own Python stack) and the job exits non-zero -- no launcher of ours involved."""
    import json
    p = _torchrun_bench({"AFSK_BENCH_FAULT": "1:die:shard built"}, ("--deadline-s", "120"))
    assert p.returncode != 0
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, (p.stdout[-1500:], p.stderr[-1500:])
    d = json.loads(lines[0])
    assert d["value"] is None and d["n_gpus"] == 2 and d["printed_by"] == "rank 0", d
    assert "SIGTERM" in d["error"] or "rank 0" in d["error"]
    assert d["heartbeats"]["1"]["phase"].startswith("shard built") and d["heartbeats"]["1"]["alive"] is False
    assert len(lines[0]) < 4096


def _ragged_rank(rank, world, port, n_total, q):
    """Two ranks on ONE device (gloo group; RCCL refuses that): unequal shards through
    dist.gather_results with the records staged through the host for the collective."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    from afskmodem_amd import batch, synth
    from afskmodem_amd import dist as adist
    from oracle import afsk_oracle as O   # checker only
    try:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(0)
        b, e = adist.shard_range(n_total, rank, world)
        n = e - b
        bf, plen, payload, ts, snr = _batch_meta(b, n)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
        x = torch.empty(n * L, dtype=torch.int16, device=dev)
        off, ln = batch.uniform_layout(n, L, dev)
        d_bf = t(bf)
        batch.modulate_batch(t(payload), t(plen), d_bf, t(ts), off, ln, L, x, True)
        q24 = np.asarray([synth.snr_to_scale_q24(s) for s in snr], np.int32)
        batch.add_noise_batch(x, off, ln, L, q24, seed=5, stream_idx_base=b)
        stride = batch.out_stride_for(L, 20)
        res = batch.demod_batch(x, off, ln, d_bf, 14000, out_stride=stride)
        torch.cuda.synchronize()
        host = batch.DemodResult(*(getattr(res, f).cpu() for f in ("bytes",) + FIELDS))
        full = adist.gather_results(host, n_total)            # CPU tensors over gloo
        # the whole batch, decoded by the oracle from the stream indices alone
        bf_a, plen_a, payload_a, ts_a, snr_a = _batch_meta(0, n_total)
        xs = O.modulate_batch(payload_a, plen_a, bf_a, ts_a, np.arange(n_total, dtype=np.int64) * L,
                              np.full(n_total, L, np.int32), n_total * L, True)
        for s_i in range(n_total):
            xs[s_i * L: (s_i + 1) * L] = O.add_noise(xs[s_i * L: (s_i + 1) * L], 5, s_i, synth.snr_to_scale_q24(snr_a[s_i]))
        want = O.demod_batch(xs, np.arange(n_total, dtype=np.int64) * L, np.full(n_total, L, np.int32), bf_a, 14000,
                             out_stride=stride)
        ok = int(full.nbytes.shape[0]) == n_total
        for f in FIELDS:
            ok = ok and bool(np.array_equal(getattr(full, f).numpy(), want[f]))
        for s_i in range(n_total):
            nb = min(int(want["nbytes"][s_i]), stride)
            ok = ok and full.bytes[s_i, :nb].numpy().tobytes() == want["bytes"][s_i, :nb].tobytes()
        q.put((rank, bool(ok), n, ""))
        dist.destroy_process_group()
    except Exception as exc:  # noqa: BLE001
        q.put((rank, False, 0, repr(exc)))
        raise


def test_unequal_shards_two_ranks_on_one_gpu():
    """n_total % world != 0 on real hardware: 2 ranks x (48 | 49) streams demodulated on device 0,
    gathered with dist.gather_results; the covering result equals the oracle's decode of all 97."""
    import torch.multiprocessing as mp
    from afskmodem_amd import _native
    assert _native.device_count() > 0, "no HIP device: GPU tests need an MI355X"
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ragged_rank, args=(r, 2, port, 97, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert sorted(g[2] for g in got) == [48, 49]
    for rank, ok, n, err in got:
        assert err == "" and ok, (rank, err)


def test_bench_three_rank_rehearsal_full_size_on_one_gpu():
    """The driver's multi-GPU scaling run, rehearsed on ONE device at FULL SIZE per rank: `python bench.py
    --gpus 3` with the default 65536 streams x 1 s per rank (3 x 6.29 GB of inputs resident at once) plus
    the config2 sub-record, reduced steps.  Launcher, 3 ranks, free port, gather_every grouping, per-rank
    gather_check, ranks_seen 3, ONE JSON line, rc 0.  (Diagnostic gloo backend: RCCL refuses several
    ranks on one device; throughput is meaningless by construction.  Three ranks, not eight: the GPU boxes
    allow 6 processes per device, and this test runner and the launching bench.py process are two of
    them; the 8-rank line recorded before that limit existed is
    profiles/archive/r3_bench_n8_diagnostic_gloo_shared_gpu.json.)"""
    import json
    import subprocess
    import sys
    import torch
    free, _ = torch.cuda.mem_get_info()
    if free < 40 * 2 ** 30:
        pytest.skip("needs ~25 GB of free HBM")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "3", "--dist-backend", "gloo",
                        "--share-gpu0", "--steps", "5", "--warmup", "1", "--preroll-ms", "0", "--min-region-ms", "0",
                        "--sub-steps", "8"],
                       capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout[-2000:]
    assert len(lines[0]) < 4096, len(lines[0])             # the N > 1 line obeys the same cap as the N = 1 line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 3 and d["ranks_seen"] == 3
    assert d["gather_check"] == [True] * 3 and d["gather_check_on_every_rank"] is True
    assert d["gather_ms"]["median"] > 0 and d["value_no_gather"] > 0
    assert d["config"]["streams_per_gpu"] == 65536 and d["config"]["streams_total"] == 3 * 65536
    assert d["config"]["workload"].startswith("configs[4]")
    assert d["roundtrip_match_rate"] == 1.0 and d["all_timed_steps_identical"] is True
    assert {"bound", "achieved", "peak", "frac", "traffic"} <= set(d["roofline"])
    assert d["sub_records"]["config2"]["gather_check"] is True and d["sub_records"]["config2"]["roundtrip"] == 1.0
    full = json.load(open(os.path.join(root, d["full_record"])))
    sub = full["sub_records"]["config2"]
    assert sub["ranks_seen"] == 3 and sub["gather_check"] == [True] * 3 and sub["roundtrip_match_rate"] == 1.0
    assert full["value"] == d["value"] and full["roofline"]["frac"] == d["roofline"]["frac"]


def test_bench_default_line_at_n1_is_compact_and_complete():
    """The driver's own invocation shape (`python bench.py --gpus 1 ...`, every default rider: config2/3/4, f1/f2/f3,
    the steady-state 36-rate table) at reduced step counts: ONE line under 4 KB carrying the contract's keys, `roofline` and
    `cpu_baseline`; the 30 KB full record in the file the line names.  (Round 3's line was 26 KB and the driver,
    which keeps ~8 KB of output, could parse nothing.)"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "2",
                        "--preroll-ms", "20", "--sub-steps", "6", "--next-reps", "3", "--wav-files", "256",
                        "--sub-cpu-sample", "64", "--cpu-sample-streams", "256", "--rates-steps", "6",
                        "--cpu-budget-s", "1"],
                       capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and len(lines[0]) < 4096, (len(lines), len(p.stdout))
    assert len(p.stderr) < 2048, p.stderr[-2000:]          # the driver's 8 KB tail holds stdout AND stderr
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "match_rate"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["config"]["workload"].startswith("configs[4]")
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(d["roofline"]) and 0 < d["roofline"]["frac"] < 1
    assert {"value", "unit", "cores", "kind", "sample"} <= set(d["cpu_baseline"]) and d["cpu_baseline"]["kind"] == "port"
    assert d["match_rate"] == 1.0 and d["roundtrip_match_rate"] == 1.0
    subs = d["sub_records"]
    assert {"config2", "config3", "config4", "config5_lead", "f1_modulate", "f2_gate", "f2_chain", "f3_wav_ingest", "rates_65536"} == set(subs)
    assert subs["f2_chain"]["roundtrip"] == 1.0 and subs["f2_chain"]["match_rate"] == 1.0 and subs["f2_chain"]["chain_ms"] > 0
    for name in ("config2", "config3", "config4", "config5_lead", "f1_modulate", "f3_wav_ingest"):
        assert subs[name]["match_rate"] == 1.0, name
    # r6: the lead-in workload (arbitrary clock index: 7 of 8 streams off the 16-byte grid) decodes and is reported
    assert subs["config5_lead"]["roundtrip"] == 1.0 and 0.85 < subs["config5_lead"]["ci_unaligned"] < 0.90
    assert "error" not in json.dumps(subs) and "incomplete" not in d
    assert subs["config4"]["ber_equals_cpu"] is True
    full = json.load(open(os.path.join(root, d["full_record"])))
    assert full["value"] == d["value"] and len(full["sub_records"]["rates_65536"]["by_baud"]) == 36
    assert full["sub_records"]["rates_65536"]["all_round_trips_exact"]


def test_bench_on_request_riders_and_a_failing_rider():
    """f5_wav_egress and rates_4096 are measured on request (--sub); and a rider that raises (here: an unwritable
    scratch directory for the file rows) becomes {"error": ...} in the line while the headline keeps its numbers."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--workload", "config2", "--steps", "6",
            "--warmup", "2", "--preroll-ms", "0", "--no-cpu-baseline", "--wav-files", "64", "--rates-steps", "4"]
    p = subprocess.run(base + ["--sub", "f5_wav_egress,rates_4096"], capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.strip()][-1])
    assert d["sub_records"]["f5_wav_egress"]["match_rate"] == 1.0 and d["sub_records"]["rates_4096"]["all_round_trips_exact"]
    p = subprocess.run(base + ["--sub", "f3_wav_ingest"], capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, AFSK_BENCH_TMPDIR="/proc/afsk_no_such_dir"))
    assert p.returncode == 0, p.stderr[-3000:]
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.strip()][-1])
    assert d["value"] > 0 and d["roundtrip_match_rate"] == 1.0 and "error" in d["sub_records"]["f3_wav_ingest"], d
