"""GPU parity (-m gpu) 1/5 -- the reference's own vectors and the API surface: every golden case through the host,
flat-host, gather-host and device entries, soft outputs, Receiver.load, the plain-C caller, error behaviour.
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


def test_golden_cases_host_entry(golden, torch_cuda):
    """Every reference-generated decode case through afsk_demod_batch_host (ragged batch)."""
    cases = golden["decode_cases"]
    xs = [build_input(c) for c in cases]
    for amp_end in sorted({c["amp_end"] for c in cases}):
        idx = [i for i, c in enumerate(cases) if c["amp_end"] == amp_end]
        res = batch.demod_host_arrays([xs[i] for i in idx],
                                      [48000 // cases[i]["baud"] for i in idx], amp_end)
        pl = res.payloads()
        for j, i in enumerate(idx):
            c = cases[i]
            assert res.clock_idx[j] == c["clock_idx"], c["tag"]
            assert res.term_frame[j] == c["term_frame"], c["tag"]
            assert res.nbits[j] == c["nbits"], c["tag"]
            assert res.nbytes[j] == c["nbytes"], c["tag"]
            assert pl[j].hex() == c["bytes_hex"], c["tag"]
            want = 1 if c["clock_idx"] == -1 else (2 if c["nbits"] == 0 else 0)
            assert res.status[j] == want, c["tag"]


def test_golden_cases_flat_host_entry_and_scratch_reuse(golden, torch_cuda):
    """afsk_demod_batch_host (one host buffer + offsets, with gaps between the streams) and
    the gather entry give the same results; the cached device scratch is reused, grown and
    released between calls without changing them."""
    cases = [c for c in golden["decode_cases"] if c["amp_end"] == 14000]
    xs = [build_input(c) for c in cases]
    bf = [48000 // c["baud"] for c in cases]
    want = batch.demod_host_arrays(xs, bf)
    gaps = [(7 * i) % 5 for i in range(len(xs))]
    ln = np.array([len(x) for x in xs], np.int32)
    off = np.cumsum([3] + [int(l) + g for l, g in zip(ln[:-1], gaps[:-1])]).astype(np.int64)
    flat = np.full(int(off[-1] + ln[-1] + 4), 999, np.int16)
    for o, x in zip(off, xs):
        flat[o: o + len(x)] = x
    for rep in range(3):
        if rep == 2:
            _native.check(_native.lib().afsk_host_scratch_release())
        sub = slice(0, len(xs) if rep != 1 else 5)          # smaller batch reuses the big scratch
        got = batch.demod_host_flat(flat, off[sub], ln[sub], bf[sub], out_stride=want.bytes.shape[1])
        for f in FIELDS:
            assert np.array_equal(getattr(got, f), getattr(want, f)[sub]), (rep, f)
        assert got.payloads() == want.payloads()[sub]
    _native.check(_native.lib().afsk_host_scratch_release())
    _native.check(_native.lib().afsk_host_scratch_release())      # idempotent
    assert batch.demod_host_arrays(xs[:3], bf[:3]).payloads() == want.payloads()[:3]


def test_gather_host_entry_large_batch_crosses_staging_windows(torch_cuda):
    """afsk_demod_streams_host with > 32 MiB of samples (several pinned windows, streams
    straddling window boundaries, an over-long stream, empty streams) equals the device path."""
    torch = torch_cuda
    b = synth_batch(torch, 900, (1200, 300, 2400), seed=5, snr_db=None)
    host = b["samples"].cpu().numpy().reshape(900, -1)
    arrays = [host[i, : 48000 - (i % 13) * 3] for i in range(900)]   # ragged, odd lengths
    arrays[17] = np.zeros(0, np.int16)
    arrays[500] = np.concatenate([host[500], host[501], host[502]])  # 144000 samples
    bf = b["h_bf"].copy()
    got = batch.demod_host_arrays(arrays, bf)
    ln = np.array([a.size for a in arrays], np.int32)
    off = np.concatenate([[0], np.cumsum(ln[:-1], dtype=np.int64)]).astype(np.int64)
    want = device_demod(torch, np.concatenate(arrays), off, ln, bf, stride=got.bytes.shape[1])
    for f in FIELDS:
        assert np.array_equal(getattr(got, f), getattr(want, f)), f
    assert got.payloads() == want.payloads()
    assert got.status[17] == _native.ST_TOO_SHORT and int((got.status == 0).sum()) >= 890


def test_host_entries_from_concurrent_threads(golden, torch_cuda):
    """The host entries run on a private non-blocking stream per calling thread: several threads
    calling them at once (independent Receivers, as in the reference, ref:275-284), while the main
    thread keeps the device busy on torch's stream, all get the results of a lone call."""
    import threading
    torch = torch_cuda
    cases = [c for c in golden["decode_cases"] if c["amp_end"] == 14000]
    xs = [build_input(c) for c in cases]
    bf = [48000 // c["baud"] for c in cases]
    want = batch.demod_host_arrays(xs, bf)
    errors = []

    def worker(tid):
        try:
            r = afskmodem.Receiver(1200)
            one = next(i for i, c in enumerate(cases) if c["baud"] == 1200 and c["nbits"] > 0)
            for it in range(12):
                sub = slice(tid % 3, len(xs) - (it % 4))
                got = (batch.demod_host_arrays(xs[sub], bf[sub]) if it % 2 == 0 else
                       batch.demod_host_flat(np.concatenate(xs[sub]),
                                             np.concatenate([[0], np.cumsum([len(x) for x in xs[sub]][:-1])]),
                                             [len(x) for x in xs[sub]], bf[sub]))
                for f in FIELDS:
                    assert np.array_equal(getattr(got, f), getattr(want, f)[sub]), (tid, it, f)
                assert got.payloads() == want.payloads()[sub], (tid, it)
                assert r.decode_frames(xs[one], string=False).hex() == cases[one]["bytes_hex"]
        except Exception as exc:  # noqa: BLE001
            errors.append(repr(exc))

    busy = torch.zeros(1 << 24, device="cuda:0")
    threads = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    for t in threads:
        t.start()
    while any(t.is_alive() for t in threads):
        busy.add_(1.0)                      # work on the caller's own stream in the meantime
        torch.cuda.synchronize()
    for t in threads:
        t.join()
    assert not errors, errors[:3]


def test_float_squelch_threshold_matches_ceil(torch_cuda):
    """ref:375 compares int(mean |x|) with the user's number: a float threshold t behaves like
    ceil(t).  A stream whose tail amplitude is exactly 14000 stops with t = 14000.5, not with 14000."""
    n, bf_v = 3, 40
    x = O.get_frames(b"threshold", 1200)[:-4800]
    tail = np.tile(np.array([14000, -14000], np.int16), 2400)      # mean |x| == 14000 per symbol
    frames = np.concatenate([x, tail, np.zeros(800, np.int16)])
    for thr, as_int in ((14000, 14000), (14000.5, 14001), (13999.2, 14000), (14001.0, 14001)):
        got = batch.demod_host_flat(frames, [0], [len(frames)], bf_v, thr)
        want = O.demod_batch(frames, np.zeros(1, np.int64), np.array([len(frames)], np.int32),
                             np.array([bf_v], np.int32), as_int, out_stride=got.bytes.shape[1])
        assert int(got.nbits[0]) == int(want["nbits"][0]), thr
        assert got.payloads()[0] == want["bytes"][0, : want["nbytes"][0]].tobytes(), thr
    lo = batch.demod_host_flat(frames, [0], [len(frames)], bf_v, 14000)
    hi = batch.demod_host_flat(frames, [0], [len(frames)], bf_v, 14000.5)
    assert int(lo.nbits[0]) > int(hi.nbits[0]) == 14 * len(b"threshold")
    r = afskmodem.Receiver(1200, amp_end_threshold=14000.5)
    assert r.decode_frames(frames, string=False) == b"threshold"


def test_plain_c_caller_decodes_on_the_gpu(torch_cuda, tmp_path):
    """tests/cabi/cabi_smoke.c: a C program (no Python, no torch, no HIP headers) synthesises a
    1200-baud stream, calls afsk_demod_batch_host and afsk_demod_streams_host, checks the bytes."""
    import subprocess
    from tests.test_host_api import build_cabi_smoke
    r = subprocess.run([build_cabi_smoke(tmp_path)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "payload 'C-ABI ok'" in r.stdout and r.stdout.strip().endswith("OK")


def test_golden_cases_device_entry(golden, torch_cuda):
    """Same cases through afsk_demod_batch on device tensors, one mixed-baud launch."""
    cases = [c for c in golden["decode_cases"] if c["amp_end"] == 14000]
    xs = [build_input(c) for c in cases]
    ln = np.array([len(x) for x in xs], np.int32)
    off = np.concatenate([[0], np.cumsum(ln[:-1], dtype=np.int64)]).astype(np.int64)
    bf = np.array([48000 // c["baud"] for c in cases], np.int32)
    res = device_demod(torch_cuda, np.concatenate(xs), off, ln, bf)
    pl = res.payloads()
    for j, c in enumerate(cases):
        got = (int(res.clock_idx[j]), int(res.term_frame[j]), int(res.nbits[j]), pl[j].hex())
        assert got == (c["clock_idx"], c["term_frame"], c["nbits"], c["bytes_hex"]), c["tag"]


def test_golden_cases_inside_large_launches(golden, torch_cuda, entry):
    """r6: the reference's own vectors through the LARGE-launch kernels -- tail hint, L2 warming, the ring re-based on
    the clock index and the ODD round forms are armed from 4096 / 6144 / 8192 (bit_frames 8: 16384) streams on, and a
    batch of a few hundred golden cases alone never gets there.  The golden cases are repeated until a launch holds 8400
    streams: all rates in one launch (per-stream and grouped entry), and -- uniform entry -- the cases of 1200 / 2400 /
    300 / 12000 / 6000 baud each in a launch of their own (6000 baud: 16500 streams).  EVERY copy must give the values
    recorded from inside the reference."""
    all_cases = [c for c in golden["decode_cases"] if c["amp_end"] == 14000]
    inputs = {c["tag"]: build_input(c) for c in all_cases}

    def run(cases, n, seed):
        order = np.tile(np.arange(len(cases)), -(-n // len(cases)))[:n]
        np.random.default_rng(seed).shuffle(order)             # neighbours in a workgroup differ in rate and length
        xs = [inputs[cases[i]["tag"]] for i in order]
        ln = np.array([len(x) for x in xs], np.int32)
        pad = (-ln) % 8                                        # streams start on 16 bytes (what upload_streams gives)
        off = np.concatenate([[0], np.cumsum((ln + pad)[:-1], dtype=np.int64)]).astype(np.int64)
        flat = np.zeros(int(off[-1] + ln[-1]), np.int16)
        for j, x in enumerate(xs):
            flat[off[j]: off[j] + ln[j]] = x
        bf = np.array([48000 // cases[i]["baud"] for i in order], np.int32)
        res = device_demod(torch_cuda, flat, off, ln, bf, stride=max(64, max(len(c["bytes_hex"]) // 2 for c in cases) + 8))
        pl = res.payloads()
        bad = []
        for j, i in enumerate(order):
            c = cases[i]
            got = (int(res.clock_idx[j]), int(res.term_frame[j]), int(res.nbits[j]), pl[j].hex())
            if got != (c["clock_idx"], c["term_frame"], c["nbits"], c["bytes_hex"]):
                bad.append((c["tag"], j, got[:3]))
        assert not bad, (len(bad), bad[:5])
        return res

    if entry != "uniform":
        res = run(all_cases, 8400, 9)
        assert len({(2 * int(c)) & 15 for c in res.clock_idx if c >= 0}) == 8      # every ring shift occurred
    else:
        for baud, n in ((1200, 8400), (2400, 8400), (300, 8400), (12000, 8400), (6000, 16500)):
            run([c for c in all_cases if c["baud"] == baud], n, baud)


def test_soft_outputs_golden_cases(golden, torch_cuda):
    """afsk_demod_batch_ex: corrected-codeword counts and per-symbol margins against the values
    recorded from inside the reference (make_golden.py, ``soft``), every decode case, one
    ragged mixed-baud launch per threshold; the hard outputs must not change."""
    import hashlib
    cases = golden["decode_cases"]
    xs = [build_input(c) for c in cases]
    for amp_end in sorted({c["amp_end"] for c in cases}):
        idx = [i for i, c in enumerate(cases) if c["amp_end"] == amp_end]
        ln = np.array([len(xs[i]) for i in idx], np.int32)
        off = np.concatenate([[0], np.cumsum(ln[:-1], dtype=np.int64)]).astype(np.int64)
        bf = np.array([48000 // cases[i]["baud"] for i in idx], np.int32)
        stride = max(160, max(cases[i]["nbytes"] for i in idx) + 8)      # multi-second cases: 100 - 900 bytes
        mstride = max(4000, max(cases[i]["soft"]["n_symbols"] for i in idx) + 8)
        res, corr, marg, nsym = soft_demod(torch_cuda, np.concatenate([xs[i] for i in idx]), off,
                                           ln, bf, amp_end, stride, mstride)
        pl = res.payloads()
        for j, i in enumerate(idx):
            c = cases[i]
            got = (int(res.clock_idx[j]), int(res.term_frame[j]), int(res.nbits[j]), pl[j].hex())
            assert got == (c["clock_idx"], c["term_frame"], c["nbits"], c["bytes_hex"]), c["tag"]
            if c["clock_idx"] < 0:
                continue
            soft = c["soft"]
            assert int(nsym[j]) == soft["n_symbols"], c["tag"]
            m = marg[j, : soft["n_symbols"]]
            assert m[:24].tolist() == soft["margins_head"], c["tag"]
            assert hashlib.sha256(m.astype("<i4").tobytes()).hexdigest() == soft["margins_sha256"], c["tag"]
            assert int(corr[j]) == soft["corrected"], c["tag"]


def test_soft_outputs_noise_vs_oracle(torch_cuda):
    """Soft outputs on noisy 1 s streams at every fast-path baud plus two generic ones; the
    margins row is compared over exactly the symbols the reference demodulated, and a narrow
    margin_stride truncates rows without touching the neighbours."""
    torch = torch_cuda
    for bauds, snrs in (((1200,), [20, 8, 5, 3, 0]), ((2400,), [12, 6, 2]), ((300,), [12, 6, 2]),
                        ((600, 4000), [10, 4]), ((300, 1200, 2400, 600), [9, 5]),
                        ((800, 500, 480, 400), [12, 6, 3]), ((480,), [9, 4]), ((400, 800), [9, 4])):
        n = 32 * len(snrs)
        b = synth_batch(torch, n, bauds, seed=77 + len(snrs) + bauds[0], snr_db=np.repeat(snrs, 32),
                        payload_len=6 if min(bauds) < 1200 else 30)
        stride = batch.out_stride_for(b["total"], int(b["h_bf"].min()))
        ms = b["total"] // int(b["h_bf"].min())
        host = b["samples"].cpu().numpy()
        want = O.demod_batch_soft(host, b["h_off"], b["h_ln"], b["h_bf"], 14000, out_stride=stride,
                                  margin_stride=ms)
        for mstride in (ms, 100):
            res = batch.demod_batch(b["samples"], b["off"], b["ln"], b["bf"], 14000,
                                    out_stride=stride, diagnostics=True, margin_stride=mstride)
            torch.cuda.synchronize()
            assert_same(res.cpu(), want, f"soft {bauds}")
            corr, marg = res.corrected.cpu().numpy(), res.margins.cpu().numpy()
            nsym = res.symbols_demodulated(b["bf"]).cpu().numpy()
            assert (nsym == want["n_symbols"]).all()
            assert (corr == want["corrected"]).all(), np.nonzero(corr != want["corrected"])[0][:8]
            col = np.arange(mstride)[None, :]
            mask = col < np.minimum(nsym, mstride)[:, None]
            bad = np.nonzero(((marg != want["margins"][:, :mstride]) & mask).any(axis=1))[0]
            assert bad.size == 0, (bauds, mstride, bad[:8])
        assert want["corrected"].max() > 0


def test_receiver_load_readme_roundtrip(golden, torch_cuda, tmp_path):
    """README.md:47-66 assertion through the drop-in API (Transmitter.save -> Receiver.load)."""
    afskmodem.LOG_LEVEL = 5
    fn = str(tmp_path / "afsk.wav")
    afskmodem.Transmitter(1200).save("Héellóo World!", fn)
    assert afskmodem.Receiver(1200).load(fn, True) == golden["readme_roundtrip"]
    # config #1: Hello World! at 1200 baud
    afskmodem.Transmitter(1200).save("Hello World!", fn)
    assert afskmodem.Receiver(1200).load(fn) == "Hello World!"
    assert afskmodem.Receiver(1200).load(fn, False) == b"Hello World!"
    # return conventions: b"" on no data even with string=True; invalid utf-8 raises
    afskmodem.Transmitter(1200).save(b"", fn)
    assert afskmodem.Receiver(1200).load(fn, True) == b""
    afskmodem.Transmitter(1200).save(b"\xff\xfe\xfd", fn)
    assert afskmodem.Receiver(1200).load(fn, False) == b"\xff\xfe\xfd"
    with pytest.raises(UnicodeDecodeError):
        afskmodem.Receiver(1200).load(fn, True)
    for baud in (300, 2400, 600):
        afskmodem.Transmitter(baud).save("Hello World!", fn)
        assert afskmodem.Receiver(baud).load(fn) == "Hello World!"
    afskmodem.LOG_LEVEL = 0


def test_invalid_baud_raises_like_reference(torch_cuda):
    x = afskmodem.Transmitter(1200).wav_samples(b"Hi!")
    with pytest.raises(Exception, match="different lengths"):
        afskmodem.Receiver(4800).decode_frames(x)
    with pytest.raises(IndexError):
        afskmodem.Receiver(20).decode_frames(x)
    assert afskmodem.Receiver(4800).decode_frames(x[:4000]) == b""   # too short: no raise
