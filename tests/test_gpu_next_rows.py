"""GPU parity (-m gpu) 5/5 -- the rows of SURVEY 8(f) and the callers either side of the path: on-device modulator and
noise, live-gate replay, .wav ingest / egress, the group plan API, Transmitter.save_batch, load_batch, gate -> demod chain.
(Split out of test_gpu_parity.py in r6; shared fixtures and helpers: tests/gpu_common.py.)"""
import os

import numpy as np
import pytest

import afskmodem_amd as afskmodem
from afskmodem_amd import _native, batch, synth
from oracle import afsk_oracle as O
from tests.golden_inputs import build_input
from tests.gpu_common import (FIELDS, REAL_DEMOD_BATCH, assert_same, device_demod, entry, soft_demod,  # noqa: F401
                              synth_batch, torch_cuda)

pytestmark = pytest.mark.gpu


def test_modulator_and_noise_match_oracle(torch_cuda):
    torch = torch_cuda
    b = synth_batch(torch, 48, (300, 1200, 2400), seed=11, snr_db=None)
    want = O.modulate_batch(b["payload"], b["plen"], b["h_bf"], b["ts"], b["h_off"], b["h_ln"],
                            48 * b["total"], True)
    got = b["samples"].cpu().numpy()
    assert np.array_equal(got, want)
    # without the wav quirk = ideal frames
    b2 = synth_batch(torch, 6, (2400,), seed=12, wav_quirk=False)
    want2 = O.modulate_batch(b2["payload"], b2["plen"], b2["h_bf"], b2["ts"], b2["h_off"],
                             b2["h_ln"], 6 * b2["total"], False)
    assert np.array_equal(b2["samples"].cpu().numpy(), want2)
    # every other valid baud, short training, ragged totals (truncation inside tones / tail)
    for total, tt in ((48000, 0.5), (9001, 0.1), (2500, 0.02)):
        b4 = synth_batch(torch, 24, (100, 600, 4000, 6000, 480, 2000, 1500, 12000), seed=15,
                         total=total, training_time=tt, payload_len=3)
        want4 = O.modulate_batch(b4["payload"], b4["plen"], b4["h_bf"], b4["ts"], b4["h_off"],
                                 b4["h_ln"], 24 * total, True)
        assert np.array_equal(b4["samples"].cpu().numpy(), want4), (total, tt)
    # noise generator: identical integers on CPU and GPU
    snr = [30, 10, 5, 0]
    b3 = synth_batch(torch, 4, (1200,), seed=13, snr_db=snr)
    clean = O.modulate_batch(b3["payload"], b3["plen"], b3["h_bf"], b3["ts"], b3["h_off"],
                             b3["h_ln"], 4 * b3["total"], True).reshape(4, -1)
    noisy = b3["samples"].cpu().numpy().reshape(4, -1)
    for s in range(4):
        assert np.array_equal(noisy[s], O.add_noise(clean[s], 14, s, int(b3["q"][s]))), s


@pytest.mark.parametrize("wav_quirk", [True, False])
def test_modulator_every_quarter_width(torch_cuda, wav_quirk):
    """Modulator vs oracle over the bit_frames the reference can transmit (divisors of 48000
    that are multiples of 4: quarter-symbol widths 1, 2, 3, 4, 5, 6 on the small-width path, 8 and
    up on the one-boundary path, up to bit_frames 2000), long training so
    most 8192-sample blocks are all-tone blocks, ragged lengths, odd 2-byte stream offsets,
    empty and long payloads."""
    torch = torch_cuda
    dev = "cuda:0"
    rng = np.random.default_rng(2024 + int(wav_quirk))
    bfs = [4, 8, 12, 16, 20, 24, 32, 40, 48, 60, 64, 80, 96, 100, 120, 128, 160, 192, 240, 300, 400,
           480, 640, 1000, 1500, 2000]
    bf = np.array([b for b in bfs for _ in range(3)], np.int32)
    n = bf.size
    ln = rng.integers(1, 60000, n).astype(np.int32)
    ln[::7] = 8192 * rng.integers(1, 6, ln[::7].size)          # exact block multiples too
    gaps = rng.integers(0, 5, n)
    off = np.concatenate([[3], 3 + np.cumsum(ln[:-1] + gaps[:-1])]).astype(np.int64)
    total = int(off[-1] + ln[-1] + 8)
    plen = rng.integers(0, 41, n).astype(np.int32)
    plen[:4] = (0, 40, 1, 0)
    ts = rng.integers(1, 3000, n).astype(np.int32)
    ts[bf > 64] = rng.integers(1, 40, int((bf > 64).sum()))
    payload = rng.integers(0, 256, (n, 40), dtype=np.uint8)
    t = lambda a: torch.from_numpy(a).to(dev)  # noqa: E731
    samples = torch.full((total,), 12345, dtype=torch.int16, device=dev)
    batch.modulate_batch(t(payload), t(plen), t(bf), t(ts), t(off), t(ln), int(ln.max()), samples,
                         wav_quirk)
    torch.cuda.synchronize()
    got = samples.cpu().numpy()
    want = O.modulate_batch(payload, plen, bf, ts, off, ln, total, wav_quirk)
    covered = np.zeros(total, bool)
    for i in range(n):
        sl = slice(int(off[i]), int(off[i] + ln[i]))
        covered[sl] = True
        bad = np.nonzero(got[sl] != want[sl])[0]
        assert bad.size == 0, (int(bf[i]), int(ln[i]), int(ts[i]), int(plen[i]), bad[:6])
    assert (got[~covered] == 12345).all()                        # nothing written outside the streams
    # bit_frames outside the domain (not a positive multiple of 4): documented all-zero stream
    bad_bf = np.array([0, 30, -8, 41], np.int32)
    o2 = (np.arange(4, dtype=np.int64) * 20000) + 1
    l2 = np.full(4, 19990, np.int32)
    s2 = torch.full((80010,), 77, dtype=torch.int16, device=dev)
    batch.modulate_batch(t(payload[:4].copy()), t(plen[:4].copy()), t(bad_bf), t(ts[:4].copy()), t(o2),
                         t(l2), 19990, s2, wav_quirk)
    torch.cuda.synchronize()
    g2 = s2.cpu().numpy()
    for i in range(4):
        assert (g2[o2[i]: o2[i] + l2[i]] == 0).all()
    assert (g2[:1] == 77).all() and (g2[o2[3] + l2[3]:] == 77).all()


def test_listen_gate_vs_reference_and_oracle(golden, torch_cuda):
    """Row f2: the block-amplitude gate on the GPU against the reference-recorded burst boundaries
    (golden) and the oracle, then gate -> demod end to end through Receiver.decode_captures."""
    from tests.golden_inputs import build_capture
    torch = torch_cuda
    for a_start, a_end in ((18000, 14000), (9000, 2500)):
        cases = [c for c in golden["listen_cases"] if c["amp_start"] == a_start]
        caps = [build_capture(c["recipe"]) for c in cases]
        samples, off, ln, max_len = batch.upload_streams(caps)
        g = batch.gate_batch(samples, off, ln, max_len, a_start, a_end, 16)
        torch.cuda.synchronize()
        nb, bs, bl, oe, amp = (t.cpu().numpy() for t in (g.n_bursts, g.burst_start, g.burst_len,
                                                          g.open_end, g.block_amp))
        # (r6) afsk_gate_batch_slots: the bursts once more, as absolute demodulator slots written by the gate kernel --
        # equal to the arithmetic burst_slots used to do with torch operations on the plain entry's outputs
        plain = batch.gate_batch(samples, off, ln, max_len, a_start, a_end, 16, slots=False)
        assert plain.slot_len is None and g.slot_len is not None
        for a_, b_ in zip(plain.burst_slots(off), g.burst_slots(off)):
            assert torch.equal(a_, b_)
        for f in ("n_bursts", "burst_start", "burst_len", "open_end", "block_amp"):
            assert torch.equal(getattr(plain, f), getattr(g, f)), f
        for i, c in enumerate(cases):
            want = [(b["start"], b["len"]) for b in c["bursts"]]
            got = [(int(bs[i, k]), int(bl[i, k])) for k in range(nb[i])]
            assert got == want and int(oe[i]) == c["open_end"], c["name"]
            assert got == O.gate_stream(caps[i], a_start, a_end, 16)[0]
            for b in range(len(caps[i]) // 2048):
                assert amp[i, b] == O.get_amplitude(caps[i][2048 * b: 2048 * b + 2048]), (c["name"], b)
        r = afskmodem.Receiver(1200, a_start, a_end)
        decoded = r.decode_captures(caps)
        for c, payloads in zip(cases, decoded):
            for p, b in zip(payloads, c["bursts"]):
                if b["len"] == b["ref_len"]:
                    assert p.hex() == b["bytes_hex"], c["name"]
    # max_bursts clamps, ragged + tiny captures, seeded random captures against the oracle
    rng = np.random.default_rng(8)
    caps = [rng.integers(-32768, 32768, int(n)).astype(np.int16) * (rng.integers(0, 2, int(n)).astype(np.int16))
            for n in (0, 100, 2048, 4096, 50000, 123457)]
    caps += [np.concatenate([rng.integers(-a, a + 1, 2048 * int(k)).astype(np.int16)
                             for a, k in zip(rng.integers(1000, 32000, 12), rng.integers(1, 4, 12))])
             for _ in range(20)]
    samples, off, ln, max_len = batch.upload_streams(caps)
    for mb in (1, 3, 16):
        g = batch.gate_batch(samples, off, ln, max_len, 18000, 14000, mb)
        torch.cuda.synchronize()
        nb, bs, bl, oe = (t.cpu().numpy() for t in (g.n_bursts, g.burst_start, g.burst_len, g.open_end))
        for i, cap in enumerate(caps):
            want, want_oe = O.gate_stream(cap, 18000, 14000, mb)
            assert [(int(bs[i, k]), int(bl[i, k])) for k in range(nb[i])] == want, (i, mb)
            assert int(oe[i]) == want_oe, (i, mb)


def test_wav_batch_ingest_and_load_batch(torch_cuda, tmp_path):
    """Row f3: many .wav files -> one device buffer -> one launch == file-by-file Receiver.load."""
    torch = torch_cuda
    rng = np.random.default_rng(12)
    names, payloads = [], []
    for i in range(40):
        data = rng.integers(32, 127, int(rng.integers(0, 30)), dtype=np.uint8).tobytes()
        fn = str(tmp_path / f"m{i}.wav")
        afskmodem.Transmitter(1200, float(rng.choice([0.1, 0.2, 0.5]))).save(data, fn)
        names.append(fn); payloads.append(data)
    samples, off, ln, max_len = batch.load_wav_batch(names)
    h = samples.cpu().numpy()
    for i, fn in enumerate(names):
        want = np.asarray(afskmodem.SoundInput.loadFromFile(fn), np.int16)
        o = int(off[i]); assert np.array_equal(h[o: o + int(ln[i])], want), i
    # the two-call form of the C-ABI (afsk_wav_probe, then afsk_wav_upload into a layout of the caller's choosing)
    import ctypes
    d_off, d_bytes, st = batch.wav_probe(names)
    assert (st == 0).all() and np.array_equal(d_bytes // 2, ln.cpu().numpy())
    lens2 = d_bytes // 2
    offs2 = np.zeros(len(names), np.int64)
    offs2[1:] = np.cumsum(((lens2 + 7) & ~7)[:-1] + 400)          # 800-byte gaps: never written
    buf = torch.full((int(offs2[-1] + lens2[-1]) + 8,), 77, dtype=torch.int16, device="cuda:0")
    arr = (ctypes.c_char_p * len(names))(*[os.fsencode(f) for f in names])
    p64 = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))  # noqa: E731
    torch.cuda.synchronize()
    _native.check(_native.lib().afsk_wav_upload(arr, p64(d_off), p64(d_bytes), p64(offs2), len(names), buf.data_ptr(), buf.numel()))
    hb = buf.cpu().numpy()
    for i, fn in enumerate(names):
        o = int(offs2[i])
        assert np.array_equal(hb[o: o + int(lens2[i])], h[int(off[i]): int(off[i]) + int(ln[i])]), i
        if i + 1 < len(names):
            assert (hb[int(offs2[i + 1]) - 300: int(offs2[i + 1])] == 77).all(), i      # the caller's gap is untouched
    afskmodem.LOG_LEVEL = 5
    r = afskmodem.Receiver(1200)
    got = r.load_batch(names)
    assert got == payloads
    assert got == [r.load(fn, False) for fn in names]
    assert r.load_batch(names[:3], string=True) == [p.decode() if p else b"" for p in payloads[:3]]
    afskmodem.LOG_LEVEL = 0


def test_native_wav_ingest_raw_riff_cases(golden, torch_cuda, tmp_path):
    """Row f3 on the device: batch.load_wav_batch (afsk_wav_probe + afsk_wav_upload: pread into
    the pinned windows, H2D) over the hand-built RIFF files equals what the reference's loader
    returned for each; a file the reference rejects raises the reference's exception."""
    from tests.test_host_api import _write_raw_cases
    from tests.golden_inputs import sha_i16
    cases = _write_raw_cases(golden, tmp_path)
    ok = [(c, fn) for c, fn in cases if c["result"] == "ok"]
    samples, off, ln, max_len = batch.load_wav_batch([fn for _, fn in ok] * 3)     # repeated: several windows' worth of files
    h, ho, hl = samples.cpu().numpy(), off.cpu().numpy(), ln.cpu().numpy()
    assert max_len == max(c["n_frames_ref"] for c, _ in ok)
    for i, (c, _) in enumerate(ok * 3):
        assert hl[i] == c["n_frames_ref"], c["name"]
        assert ho[i] % 8 == 0
        assert sha_i16(h[ho[i]: ho[i] + hl[i]]) == c["frames_sha256"], c["name"]
    for c, fn in cases:
        if c["result"] != "ok":
            with pytest.raises(BaseException) as ei:
                batch.load_wav_batch([ok[0][1], fn, ok[1][1]])
            assert type(ei.value).__name__ == c["exc_type"] and str(ei.value) == c["exc_msg"], c["name"]
    # a big batch crossing the 32 MiB staging windows: 700 x 1 s files
    t = afskmodem.Transmitter(1200)
    names = []
    for i in range(700):
        fn = str(tmp_path / f"big{i}.wav")
        if i < 8:
            t.save(bytes([65 + i]) * (20 + i), fn)
        else:
            import shutil
            shutil.copyfile(str(tmp_path / f"big{i % 8}.wav"), fn)
        names.append(fn)
    got = afskmodem.Receiver(1200).load_batch(names, string=False)
    assert got == [bytes([65 + (i % 8)]) * (20 + (i % 8)) for i in range(700)]


def test_plan_freed_inside_a_stream_capture_is_parked_not_synchronised(torch_cuda, entry):
    """r5 advisor: the last reference to a cached plan may go away INSIDE a HIP stream capture (rebinding the result of
    a demod_batch call while capturing).  GroupPlan.__del__ must not synchronise the device there -- that would
    invalidate the capture -- but park the handle; the next plan construction (or release_parked_plans) frees it."""
    if entry != "grouped":
        pytest.skip("entry-independent: runs once")
    torch = torch_cuda
    b = synth_batch(torch, 64, (1200, 300, 2400, 600, 800), seed=77)
    stride = batch.out_stride_for(48000, int(b["h_bf"].min()))
    res = REAL_DEMOD_BATCH(b["samples"], b["off"], b["ln"], b["h_bf"], 14000, out_stride=stride, entry="grouped")
    torch.cuda.synchronize()
    want = res.cpu()
    with batch._PLAN_CACHE_LOCK:
        batch._PLAN_CACHE.clear()                 # the result now holds the only reference to its plan
    out = batch.alloc_result(64, stride, "cuda:0")
    plan = batch.GroupPlan(b["h_bf"], "cuda:0")
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    before = len(batch._PARKED_PLANS)
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            REAL_DEMOD_BATCH(b["samples"], b["off"], b["ln"], None, 14000, out=out, plan=plan)
            del res                               # -> GroupPlan.__del__ while capturing
    assert len(batch._PARKED_PLANS) == before + 1
    graph.replay()                                # the capture survived
    torch.cuda.synchronize()
    got = out.cpu()
    for f in FIELDS:
        assert np.array_equal(getattr(got, f), getattr(want, f)), f
    assert batch.release_parked_plans() == 0


def test_grouped_dispatch_plan_api(torch_cuda, entry):
    """afsk_group_plan_* / afsk_demod_batch_grouped directly (not through the `entry` fixture): bucket order and
    counts, a plan reused across launches and thresholds, status 3 for streams whose host-side bit_frames is
    invalid (what the per-stream kernel writes for them), soft outputs at the original stream numbers, a
    plan/batch mismatch refused, and equality with the per-stream entry on the same batch, field by field."""
    if entry != "grouped":
        pytest.skip("entry-independent: runs once")
    torch = torch_cuda
    n = 600
    b = synth_batch(torch, n, (1200, 375, 300, 96, 2400, 160), seed=606, snr_db=np.where(np.arange(n) % 4 == 0, 7.0, 40.0))
    bf_h = b["h_bf"].copy()
    bf_h[[11, 222]] = (42, 0)                              # not a multiple of 4 / zero: refused, status 3
    bf_h[333] = 2048                                       # 2 * bf >= 4096
    plan = batch.GroupPlan(bf_h)
    groups = plan.groups()
    assert sum(c for _, c in groups) == n and groups[-1] == (0, 3)
    counts = [c for _, c in groups[:-1]]
    assert counts == sorted(counts, reverse=True) and {g for g, _ in groups[:-1]} == {40, 128, 160, 500, 20, 300}
    stride = batch.out_stride_for(48000, 20)
    ms = 48000 // 20
    for amp_end in (14000, 9000.5):
        got = REAL_DEMOD_BATCH(b["samples"], b["off"], b["ln"], None, amp_end, out_stride=stride, plan=plan,
                               diagnostics=True, margin_stride=ms)
        ref = REAL_DEMOD_BATCH(b["samples"], b["off"], b["ln"], torch.from_numpy(bf_h).to("cuda:0"), amp_end,
                               out_stride=stride, validate=False, entry="mixed", diagnostics=True, margin_stride=ms)
        torch.cuda.synchronize()
        g, r = got.cpu(), ref.cpu()
        for f in FIELDS:
            assert np.array_equal(getattr(g, f), getattr(r, f)), (f, amp_end)
        assert (g.status[[11, 222, 333]] == _native.ST_INVALID_BAUD).all()
        m = np.arange(stride)[None, :] < np.minimum(r.nbytes, stride)[:, None]
        assert not ((g.bytes != r.bytes) & m).any()
        assert torch.equal(got.corrected, ref.corrected)
        ok = np.setdiff1d(np.arange(n), [11, 222, 333])
        nsym = got.symbols_demodulated(torch.from_numpy(np.maximum(bf_h, 4)).to("cuda:0")).cpu().numpy()
        gm, rm = got.margins.cpu().numpy(), ref.margins.cpu().numpy()
        for s_i in ok[::7]:
            k = min(int(nsym[s_i]), ms)
            assert np.array_equal(gm[s_i, :k], rm[s_i, :k]), s_i
    with pytest.raises(ValueError, match="plan covers"):
        REAL_DEMOD_BATCH(b["samples"], b["off"][:10].contiguous(), b["ln"][:10].contiguous(), None, 14000,
                         out_stride=stride, plan=plan)
    with pytest.raises(ValueError, match="1 or 600"):
        REAL_DEMOD_BATCH(b["samples"], b["off"], b["ln"], [40, 40, 40], 14000, out_stride=stride)
    plan.close()
    with pytest.raises(ValueError, match="closed"):
        REAL_DEMOD_BATCH(b["samples"], b["off"], b["ln"], None, 14000, out_stride=stride, plan=plan)
    # auto: a host list with several rates takes the grouped dispatch, an all-equal one the uniform kernel
    auto = REAL_DEMOD_BATCH(b["samples"], b["off"], b["ln"], [int(v) for v in b["h_bf"]], 14000, out_stride=stride)
    torch.cuda.synchronize()
    want = O.demod_batch(b["samples"].cpu().numpy(), b["h_off"], b["h_ln"], b["h_bf"], 14000, out_stride=stride, n_threads=16)
    assert_same(auto.cpu(), want, "auto -> grouped")
    # host entries with several rates: the same dispatch behind afsk_demod_batch_host / afsk_demod_streams_host
    h = b["samples"].cpu().numpy()
    flat = batch.demod_host_flat(h, b["h_off"], b["h_ln"], b["h_bf"], 14000, stride)
    assert_same(flat, want, "host flat -> grouped")
    arrs = batch.demod_host_arrays([h[i * 48000: (i + 1) * 48000] for i in range(n)], b["h_bf"], 14000)
    for f in FIELDS:
        assert np.array_equal(getattr(arrs, f), want[f]), f


@pytest.mark.filterwarnings("ignore::pytest.PytestUnraisableExceptionWarning")   # CPython 3.10's Wave_write.__del__ after a failed open
def test_transmitter_save_batch_writes_the_reference_files(golden, torch_cuda, tmp_path, entry):
    """Transmitter.save for many payloads (device modulator + afsk_wav_egress): every file equals, byte for byte,
    what Transmitter.save writes on the host -- which the reference's own digests pin (the 72 frame / wav cases
    of the fixture: payload samples by SHA-256; the README file by its whole-file digest) -- and decodes back
    through Receiver.load_batch.  Also a baud rate the device modulator has no geometry for (host fallback) and a
    batch larger than one staging window."""
    import hashlib
    import wave
    if entry != "uniform":
        pytest.skip("entry-independent: runs once")
    afskmodem.LOG_LEVEL = 5
    by_tx = {}
    for c in golden["frames"]:
        by_tx.setdefault((c["baud"], c["training_time"]), []).append(c)
    for (baud, tt), cases in by_tx.items():
        t = afskmodem.Transmitter(baud, tt)
        names = [str(tmp_path / f"g_{baud}_{tt}_{i}.wav") for i in range(len(cases))]
        t.save_batch([bytes.fromhex(c["payload_hex"]) for c in cases], names)
        for c, fn in zip(cases, names):
            with wave.open(fn, "rb") as f:
                assert (f.getnchannels(), f.getsampwidth(), f.getframerate()) == (1, 2, 48000)
                raw = f.readframes(f.getnframes())
            assert len(raw) // 2 == c["n_wav"] and hashlib.sha256(raw).hexdigest() == c["wav_sha256"], (baud, tt, c["payload"])
            host = tmp_path / "host.wav"
            t.save(bytes.fromhex(c["payload_hex"]), str(host))
            assert open(fn, "rb").read() == host.read_bytes(), (baud, tt, c["payload"])
    readme = tmp_path / "readme.wav"
    afskmodem.Transmitter(1200).save_batch(["Héellóo World!"], [str(readme)])   # the README payload: str -> utf-8 (ref:482-483)
    assert hashlib.sha256(readme.read_bytes()).hexdigest() == golden["readme_wav_file_sha256"]
    # 600 x 1 s payloads (57 MB: several staging windows), decoded back
    t = afskmodem.Transmitter(1200)
    payloads = [bytes([65 + i % 26]) * 34 for i in range(600)]
    names = [str(tmp_path / f"b{i:03d}.wav") for i in range(600)]
    t.save_batch(payloads, names)
    assert afskmodem.Receiver(1200).load_batch(names) == payloads
    # 48000 / 8000 = 6 is not a multiple of 4 (mark tone of 4, space tone of 6 samples): the reference still modulates
    # it, the device modulator has no geometry for it -> host path
    odd = afskmodem.Transmitter(8000, 0.1)
    odd.save_batch([b"xy"], [str(tmp_path / "odd.wav")])
    host = tmp_path / "odd_host.wav"
    odd.save(b"xy", str(host))
    assert (tmp_path / "odd.wav").read_bytes() == host.read_bytes()
    with pytest.raises(ValueError):
        t.save_batch([b"a", b"b"], [str(tmp_path / "one.wav")])
    with pytest.raises(FileNotFoundError):
        t.save_batch([b"a"], [str(tmp_path / "nope" / "x.wav")])


def test_load_batch_for_several_receivers(torch_cuda, tmp_path, entry):
    """afskmodem_amd.load_batch: files of five baud rates and two squelch thresholds decoded on behalf of their own
    Receivers in one ingest + one grouped launch per threshold; every payload equals what that Receiver's own
    load() returns for the file."""
    if entry != "grouped":
        pytest.skip("entry-independent: runs once")
    afskmodem.LOG_LEVEL = 5
    rng = np.random.default_rng(77)
    rx = {b: afskmodem.Receiver(b) for b in (300, 1200, 2400, 375, 96)}
    rx_hi = afskmodem.Receiver(1200, amp_end_threshold=20000)
    receivers, names, want = [], [], []
    for i in range(60):
        baud = (300, 1200, 2400, 375, 96)[i % 5]
        data = rng.integers(0, 256, int(rng.integers(0, 6)), dtype=np.uint8).tobytes()
        fn = str(tmp_path / f"m{i:02d}.wav")
        afskmodem.Transmitter(baud, 0.1).save(data, fn)
        r = rx_hi if (baud == 1200 and i % 2) else rx[baud]
        receivers.append(r); names.append(fn); want.append(data)
    got = afskmodem.load_batch(receivers, names)
    assert got == want
    assert got == [r.load(fn, string=False) for r, fn in zip(receivers, names)]
    with pytest.raises(ValueError):
        afskmodem.load_batch(receivers[:3], names[:2])


def test_gate_to_demod_chain_without_host_sync_and_as_a_graph(golden, torch_cuda, entry):
    """f2 -> demod with no host round trip: gate_batch -> GateResult.burst_slots (fixed slots, length 0 where a capture
    has fewer bursts) -> demod_batch gives, slot by slot, what the compacting route (burst_streams: a nonzero, i.e. a
    synchronisation) gives burst by burst -- which the reference-recorded listen cases pin -- and the whole chain
    is captured into ONE HIP graph and replayed on new captures in the same buffer."""
    from tests.golden_inputs import build_capture
    if entry != "uniform":
        pytest.skip("entry-independent: runs once")
    torch = torch_cuda
    cases = [c for c in golden["listen_cases"] if c["amp_start"] == 18000]
    caps = [build_capture(c["recipe"]) for c in cases]
    samples, off, ln, max_len = batch.upload_streams(caps)
    mb = 4
    stride = batch.out_stride_for(max_len, 40)

    def chain(out=None):
        g = batch.gate_batch(samples, off, ln, max_len, 18000, 14000, mb)
        s_off, s_len = g.burst_slots(off)
        return g, batch.demod_batch(samples, s_off, s_len, 40, 14000, out=out, out_stride=None if out is not None else stride)

    g, res = chain()
    torch.cuda.synchronize()
    owner, b_off, b_len = g.burst_streams(off)
    ref = batch.demod_batch(samples, b_off, b_len, 40, 14000, out_stride=stride)
    torch.cuda.synchronize()
    nb = g.n_bursts.cpu().numpy()
    slots, refp = res.cpu(), ref.payloads()
    k = 0
    for s_i in range(len(caps)):
        for j in range(mb):
            slot = s_i * mb + j
            if j < nb[s_i]:
                assert slots.payloads()[slot] == refp[k] and slots.status[slot] == ref.status[k].item(), (s_i, j)
                k += 1
            else:
                assert slots.status[slot] == _native.ST_TOO_SHORT and slots.nbytes[slot] == 0
    assert k == len(refp) and k > 0
    for c, s_i in zip(cases, range(len(caps))):                    # and the reference's own bursts
        for j, b in enumerate(c["bursts"][: mb]):
            if b["len"] == b["ref_len"]:
                assert slots.payloads()[s_i * mb + j].hex() == b["bytes_hex"], c["name"]
    # the chain as one graph, replayed after the captures were swapped for others (same layout)
    out = batch.alloc_result(len(caps) * mb, stride, "cuda:0")
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            chain(out)
    out.flat.zero_()
    graph.replay()
    torch.cuda.synchronize()
    got = out.cpu()
    assert got.payloads() == slots.payloads() and np.array_equal(got.status, slots.status)
    samples.zero_()                                                  # other data, same graph: silence -> no bursts at all
    graph.replay()
    torch.cuda.synchronize()
    assert (out.cpu().status == _native.ST_TOO_SHORT).all()
