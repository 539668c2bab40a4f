"""GPU parity (-m gpu) 2/5 -- the HIP path against the CPU oracle on seeded inputs: clean and noisy batches, ragged and
edge lengths, thresholds, the maximum stream length, garbage, fuzz, run-time geometries, every accepted bit_frames,
device-side guards, graph capture and side streams.
(Split out of test_gpu_parity.py in r6; shared fixtures and helpers: tests/gpu_common.py.)"""
import os

import numpy as np
import pytest

import afskmodem_amd as afskmodem
from afskmodem_amd import _native, batch, synth
from oracle import afsk_oracle as O
from tests.golden_inputs import build_input
from tests.gpu_common import (FIELDS, REAL_DEMOD_BATCH, assert_same, device_demod, entry, large_launch_streams,  # noqa: F401
                              soft_demod, synth_batch, torch_cuda)

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,bauds", [(512, (1200,)), (384, (300, 1200, 2400)), (512, (600,)), (256, (600, 1200, 300, 2400)),
                                     (192, (12000, 6000, 4000, 3000, 2000, 1500, 1000, 750)),
                                     (128, (12000,)), (128, (750,)), (120, (800, 500, 480, 400, 375, 250)), (96, (600, 100, 4000, 6000, 480, 2000))])
def test_clean_batches_vs_oracle(torch_cuda, n, bauds):
    """Config #2 / #3 shapes at test size: every output equals the CPU oracle's, and the
    decoded payload equals what was modulated (round trip)."""
    torch = torch_cuda
    pl = None if all(b in synth.ONE_SECOND_PAYLOAD for b in bauds) else 3
    # the .wav writer's decimate/duplicate quirk (ref:239-244) destroys a 12000-baud mark tone
    # (quarter symbol = one frame), in the reference too: the pure 12000-baud batch uses ideal frames
    quirk = bauds != (12000,)
    b = synth_batch(torch, n, bauds, seed=21, payload_len=pl, wav_quirk=quirk)
    stride = batch.out_stride_for(b["total"], int(b["h_bf"].min()))
    res = batch.demod_batch(b["samples"], b["off"], b["ln"], b["h_bf"], 14000, out_stride=stride)
    torch.cuda.synchronize()
    got = res.cpu()
    want = O.demod_batch(b["samples"].cpu().numpy(), b["h_off"], b["h_ln"], b["h_bf"], 14000,
                         out_stride=stride, n_threads=8)
    assert_same(got, want, f"clean {bauds}")
    for s, data in enumerate(got.payloads()):
        if quirk and b["h_bf"][s] == 4:
            continue                      # not decodable after the quirk (GPU == oracle checked above)
        assert data == b["payload"][s, : b["plen"][s]].tobytes(), s


def test_noise_sweep_vs_oracle(torch_cuda):
    """Config #4 shape at test size: SNR 30 -> 0 dB, GPU and CPU results coincide exactly,
    including squelch over-read and false terminators at low SNR."""
    torch = torch_cuda
    snrs = [30, 25, 20, 15, 10, 7, 5, 3, 0]
    n = 64 * len(snrs)
    snr = np.repeat(snrs, 64)
    b = synth_batch(torch, n, (1200,), seed=31, snr_db=snr)
    stride = batch.out_stride_for(b["total"], 40)
    got = batch.demod_batch(b["samples"], b["off"], b["ln"], 40, 14000, out_stride=stride).cpu()
    want = O.demod_batch(b["samples"].cpu().numpy(), b["h_off"], b["h_ln"], b["h_bf"], 14000,
                         out_stride=stride, n_threads=8)
    assert_same(got, want, "noise sweep")
    # the sweep must actually exercise the hard cases
    assert (got.clock_idx != 0).any() and (got.nbits > 476).any()
    ber_ok = [all(got.payloads()[i][:34] == b["payload"][i, :34].tobytes()
                  for i in range(k * 64, k * 64 + 64)) for k in range(len(snrs))]
    assert ber_ok[0] and ber_ok[4]          # 30 dB and 10 dB decode error-free
    for baud, bf in ((300, 160), (2400, 20), (600, 80), (800, 60), (500, 96), (480, 100), (400, 120)):
        bb = synth_batch(torch, 96, (baud,), seed=32 + bf, snr_db=np.repeat([12, 6, 2], 32),
                         payload_len=None if baud in synth.ONE_SECOND_PAYLOAD else 10)
        st = batch.out_stride_for(bb["total"], bf)
        g = batch.demod_batch(bb["samples"], bb["off"], bb["ln"], bf, 14000, out_stride=st).cpu()
        w = O.demod_batch(bb["samples"].cpu().numpy(), bb["h_off"], bb["h_ln"], bb["h_bf"], 14000,
                          out_stride=st, n_threads=8)
        assert_same(g, w, f"noise {baud}")


def test_ragged_unaligned_and_edge_lengths(torch_cuda):
    """Ragged lengths, odd sample offsets (2-byte aligned streams), the 4096 boundary,
    streams with no tail silence, empty batch, truncating out_stride."""
    torch = torch_cuda
    rng = np.random.default_rng(5)
    tx = afskmodem.Transmitter(1200, 0.1)
    pieces, bf = [], []
    lens_wanted = [0, 1, 4000, 4095, 4096, 4097, 4136, 4137, 5000, 9999, 12345, 20001]
    for i, L in enumerate(lens_wanted):
        w = tx.wav_samples(rng.integers(0, 256, 5, dtype=np.uint8).tobytes())
        lead = rng.integers(0, 300)
        w = np.concatenate([np.zeros(lead, np.int16), w])
        pieces.append(w[:L] if L <= len(w) else np.concatenate([w, np.zeros(L - len(w), np.int16)]))
        bf.append(40)
    for baud in (300, 2400, 600):      # no tail: final symbol ends at the buffer end
        t = afskmodem.Transmitter(baud, 0.1)
        fr = t.frames(b"xyz")[:-4800]
        for extra in (0, 1, 2, 3, 5):
            pieces.append(np.concatenate([fr, np.zeros(extra, np.int16)]))
            bf.append(48000 // baud)
    # odd gaps between streams so that bases are only 2-byte aligned
    gaps = [1, 3, 0, 7, 1, 1, 5, 0, 9, 1, 3, 1] + [1] * (len(pieces) - 12)
    flat, off = [], []
    pos = 0
    for p, g in zip(pieces, gaps):
        flat.append(rng.integers(-30000, 30000, g).astype(np.int16)); pos += g
        off.append(pos); flat.append(p); pos += len(p)
    flat.append(np.zeros(3, np.int16))
    flat = np.concatenate(flat)
    ln = np.array([len(p) for p in pieces], np.int32)
    off = np.array(off, np.int64)
    bf = np.array(bf, np.int32)
    got = device_demod(torch, flat, off, ln, bf, stride=64)
    want = O.demod_batch(flat, off, ln, bf, 14000, out_stride=64)
    assert_same(got, want, "ragged")
    assert (got.status == 1).sum() == 4 and (got.status == 0).sum() >= 10
    # truncating stride: nbytes still reports the full count, row holds the prefix
    got2 = device_demod(torch, flat, off, ln, bf, stride=4)
    assert np.array_equal(got2.nbytes, want["nbytes"])
    n4 = np.minimum(want["nbytes"], 4)
    for s in range(len(ln)):
        assert got2.bytes[s, : n4[s]].tobytes() == want["bytes"][s, : n4[s]].tobytes()
    # empty batch
    assert batch.demod_host_arrays([], 40).nbytes.size == 0


def test_squelch_thresholds_and_long_stream(torch_cuda):
    torch = torch_cuda
    b = synth_batch(torch, 32, (1200,), seed=41, snr_db=np.repeat([40, 8], 16))
    h = b["samples"].cpu().numpy()
    for amp_end in (0, -5, 1, 14000, 20000, 32767, 32768, 40000, 100000):
        stride = 400
        got = batch.demod_batch(b["samples"], b["off"], b["ln"], 40, amp_end, out_stride=stride).cpu()
        want = O.demod_batch(h, b["h_off"], b["h_ln"], b["h_bf"], amp_end, out_stride=stride, n_threads=8)
        assert_same(got, want, f"amp_end {amp_end}")
    # one long stream (20 s, 2400 baud, 1500-byte payload): many DMA rounds, many bytes
    t = afskmodem.Transmitter(2400, 0.5)
    data = np.random.default_rng(3).integers(0, 256, 1500, dtype=np.uint8).tobytes()
    w = t.wav_samples(data)
    got = batch.demod_host_arrays([w, w[: len(w) // 2]], 20)
    want = O.demod_batch(np.concatenate([w, w[: len(w) // 2]]), [0, len(w)],
                         [len(w), len(w) // 2], [20, 20], 14000, out_stride=got.bytes.shape[1])
    assert_same(got, want, "long")
    assert got.payloads()[0] == data


_MAXLEN_ORACLE = {}


@pytest.mark.parametrize("bf", [40, 300])
def test_maximum_stream_length(torch_cuda, entry, bf):
    """The longest stream the C-ABI accepts (AFSK_MAX_STREAM_LEN = 2^30 - 2^15 samples, 6.2 hours of audio:
    byte offsets just below 2^31), filled to the end with a Transmitter frame carrying 1.9 MB (1200 baud) /
    0.26 MB (160 baud) of payload, inside a launch large enough to arm the tail hint (the other streams are
    empty): hundreds of thousands of ring laps and deferred ECC flushes, the 32-bit position arithmetic at
    its limit, the probe spacing of the hint at its largest.  Every output equals the CPU oracle's."""
    torch = torch_cuda
    free, _ = torch.cuda.mem_get_info()
    if free < 12 * 2 ** 30:
        pytest.skip("needs ~6 GB of free HBM")
    dev = "cuda:0"
    L = _native.MAX_STREAM_LEN
    baud = 48000 // bf
    ts = synth.ts_cycles_for(baud, 0.5)
    plen = (L - ts * 2 * bf - 4 * bf - 4800) // (14 * bf)
    payload = synth.payload_bytes(1234 + bf, 0, 1, plen)
    n = 6300                                             # >= kHintMinStreams (mixed) and kHintMinStreamsUniform
    off = np.zeros(n, np.int64)
    ln = np.zeros(n, np.int32)
    ln[17] = L                                           # stream 17 is the long one, at sample offset 0
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    x = torch.zeros(L, dtype=torch.int16, device=dev)
    batch.modulate_batch(t(payload), t(np.array([plen], np.int32)), t(np.array([bf], np.int32)),
                         t(np.array([ts], np.int32)), t(off[:1]), t(ln[17:18]), L, x, True)
    torch.cuda.synchronize()
    stride = (plen + 8) & ~3
    if bf not in _MAXLEN_ORACLE:                         # the oracle needs ~10 s per stream: once for both entries
        w = O.demod_batch(x.cpu().numpy(), off[:1], ln[17:18], np.array([bf], np.int32), 14000, out_stride=stride)
        _MAXLEN_ORACLE[bf] = w
    want = _MAXLEN_ORACLE[bf]
    res = batch.demod_batch(x, t(off), t(ln), np.full(n, bf, np.int32), 14000, out_stride=stride)
    torch.cuda.synchronize()
    got_nb, got_bits = int(res.nbytes[17].item()), int(res.nbits[17].item())
    assert (got_nb, got_bits) == (int(want["nbytes"][0]), int(want["nbits"][0])) and got_nb == plen
    assert int(res.clock_idx[17].item()) == int(want["clock_idx"][0]) == 0
    assert int(res.term_frame[17].item()) == int(want["term_frame"][0])
    assert int(res.status[17].item()) == 0
    row = res.bytes[17, :plen].cpu().numpy()
    assert np.array_equal(row, want["bytes"][0, :plen]) and np.array_equal(row, payload[0])
    st = res.status.cpu().numpy()
    assert (np.delete(st, 17) == _native.ST_TOO_SHORT).all()
    del x, res
    torch.cuda.empty_cache()


def test_runtime_geometry_rates_vs_oracle(torch_cuda):
    """Every valid bit_frames above 120 except 160 (375 baud and below) runs the single-pass ring
    with a geometry computed at run time (lanes per symbol, symbols per round, run-time clock
    recovery): clean, noisy and offset streams, soft outputs included, against the oracle."""
    torch = torch_cuda
    rng = np.random.default_rng(808)
    bauds = (375, 250, 240, 200, 160, 150, 125, 120, 100, 96, 80, 75, 60, 50, 48, 40, 32, 30, 25, 24)
    pieces, bfs, clean_payload = [], [], []
    for baud in bauds:
        bf = 48000 // baud
        tx = afskmodem.Transmitter(baud, max(0.05, 24.0 / baud))
        for k in range(6):
            data = rng.integers(0, 256, 2 + (k % 3), dtype=np.uint8).tobytes()
            w = tx.wav_samples(data)
            lead = 0 if k == 0 else int(rng.integers(1, 4 * bf))
            x = np.concatenate([rng.integers(-300, 300, lead).astype(np.int16), w])
            if k == 3:
                x = x[: len(x) - 4800]                       # no tail silence: last-symbol rule (i < len - bf)
            if k >= 4:
                x = np.clip(x.astype(np.int32) + rng.normal(0, 9000 if k == 4 else 20000, len(x)), -32768, 32767).astype(np.int16)
            pieces.append(x); bfs.append(bf); clean_payload.append(data if k == 0 else None)
        pieces.append(rng.integers(-32768, 32768, 9000).astype(np.int16)); bfs.append(bf); clean_payload.append(None)
        pieces.append(np.zeros(5000, np.int16)); bfs.append(bf); clean_payload.append(None)
        if baud in (375, 250, 200, 160, 100):
            # long payloads: many ring laps and several deferred 64-byte Hamming flushes
            data = rng.integers(0, 256, 150 + baud % 7, dtype=np.uint8).tobytes()
            w = afskmodem.Transmitter(baud, 0.1).wav_samples(data)
            pieces.append(np.concatenate([np.zeros(3, np.int16), w])); bfs.append(bf); clean_payload.append(data)
    ln = np.array([len(p) for p in pieces], np.int32)
    off = np.concatenate([[0], np.cumsum(ln[:-1], dtype=np.int64)]).astype(np.int64)
    bf = np.array(bfs, np.int32)
    flat = np.concatenate(pieces)
    stride, ms = 192, 2400
    want = O.demod_batch_soft(flat, off, ln, bf, 14000, out_stride=stride, margin_stride=ms)
    got, corr, marg, nsym = soft_demod(torch, flat, off, ln, bf, 14000, stride, ms)
    assert_same(got, want, "run-time geometry")
    assert (nsym == want["n_symbols"]).all()
    assert (corr == want["corrected"]).all()
    col = np.arange(ms)[None, :]
    mask = col < np.minimum(nsym, ms)[:, None]
    bad = np.nonzero(((marg != want["margins"]) & mask).any(axis=1))[0]
    assert bad.size == 0, bad[:8]
    pl = got.payloads()
    for i, data in enumerate(clean_payload):         # the clean, offset-free stream of every rate round-trips
        if data is not None:
            assert pl[i] == data, (i, int(bf[i]))
    assert sum(n > 0 for n in got.nbytes) > len(pieces) // 2
    for amp_end in (0, 22000):
        g = device_demod(torch, flat, off, ln, bf, amp_end=amp_end, stride=stride)
        w = O.demod_batch(flat, off, ln, bf, amp_end, out_stride=stride, n_threads=16)
        assert_same(g, w, f"run-time geometry amp_end {amp_end}")


def test_every_bit_frames_value_the_kernel_accepts(torch_cuda):
    """The device entry takes bit_frames as a device array and accepts every multiple of 4 with
    2 * bf < 4096 (host wrappers additionally require 48000 % bf == 0, like the reference): ALL 511
    values, 4 ... 2044 -- the compile-time geometries and, for everything else, the run-time one --
    on a clean stream and a noisy one each, against the oracle (which is a literal scalar loop for
    any bf)."""
    torch = torch_cuda
    dev = "cuda:0"
    rng = np.random.default_rng(4044)
    bfs = np.arange(4, 2048, 4, dtype=np.int32)
    n = 2 * len(bfs)
    bf = np.repeat(bfs, 2)
    ts = np.where(bf <= 64, 60, np.where(bf <= 320, 12, 4)).astype(np.int32)
    plen = np.full(n, 2, np.int32)
    payload = rng.integers(0, 256, (n, 2), dtype=np.uint8)
    ln = (ts * 2 * bf + 4 * bf + 28 * bf + 4800).astype(np.int32)
    ln = np.maximum(ln, 4200).astype(np.int32)
    off = np.concatenate([[0], np.cumsum(ln[:-1], dtype=np.int64)]).astype(np.int64)
    total = int(off[-1] + ln[-1])
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    x = torch.zeros(total, dtype=torch.int16, device=dev)
    d_off, d_ln, d_bf = t(off), t(ln), t(bf)
    batch.modulate_batch(t(payload), t(plen), d_bf, t(ts), d_off, d_ln, int(ln.max()), x, False)
    q = np.where(np.arange(n) % 2 == 0, synth.snr_to_scale_q24(60.0), synth.snr_to_scale_q24(8.0)).astype(np.int32)
    batch.add_noise_batch(x, d_off, d_ln, int(ln.max()), q, seed=77)
    stride = 16
    res = batch.demod_batch(x, d_off, d_ln, d_bf, 14000, out_stride=stride, validate=False)
    torch.cuda.synchronize()
    got = res.cpu()
    want = O.demod_batch(x.cpu().numpy(), off, ln, bf, 14000, out_stride=stride, n_threads=16)
    assert_same(got, want, "every bit_frames")
    ok = sum(got.payloads()[i] == payload[i].tobytes() for i in range(0, n, 2))
    assert ok > 0.9 * len(bfs), ok        # the clean stream of (nearly) every width round-trips


def test_random_garbage_streams(torch_cuda):
    """Uniform full-range int16 garbage, constant extremes, and alternating full-scale values:
    no training sequence, many false terminators, -32768 everywhere -- the integer paths
    (abs(-32768) = 32768, limiter dead zone edges, SAD sums at their maxima) must still agree."""
    torch = torch_cuda
    rng = np.random.default_rng(2718)
    pieces, bfs = [], []
    for bf in (20, 40, 160, 80, 480, 8, 60, 96, 100, 120):
        for L in (4096, 6000, 20000):
            pieces.append(rng.integers(-32768, 32768, L).astype(np.int16)); bfs.append(bf)
        pieces.append(np.full(9000, -32768, np.int16)); bfs.append(bf)
        pieces.append(np.full(9000, 32767, np.int16)); bfs.append(bf)
        pieces.append(np.tile(np.array([-32768, 32767], np.int16), 5000)); bfs.append(bf)
        pieces.append(np.tile(np.array([512, -512, 513, -513, 0], np.int16), 2000)); bfs.append(bf)
        sq = np.repeat(np.tile(np.array([32767, -32768], np.int16), 40000 // bf), bf // 2)
        pieces.append(sq[:30000]); bfs.append(bf)            # a pure space tone: no terminator
    ln = np.array([len(p) for p in pieces], np.int32)
    off = np.concatenate([[0], np.cumsum(ln[:-1], dtype=np.int64)]).astype(np.int64)
    bf = np.array(bfs, np.int32)
    flat = np.concatenate(pieces)
    for amp_end in (14000, 0):
        got = device_demod(torch, flat, off, ln, bf, amp_end=amp_end, stride=512)
        want = O.demod_batch(flat, off, ln, bf, amp_end, out_stride=512, n_threads=8)
        assert_same(got, want, f"garbage amp_end={amp_end}")
    assert (got.nbytes > 0).any()


def test_fuzz_noise_streams_every_rate(torch_cuda):
    """A reduced tools/fuzz_gpu.py inside the suite: for each of the 17 rates, 400 streams of
    band-limited garbage + spliced short bursts at random offsets (uniformly distributed clock
    indices, chance terminators, squelch stops anywhere), three squelch thresholds -- every output
    equals the CPU oracle's."""
    torch = torch_cuda
    rng = np.random.default_rng(20261003)
    for baud in (300, 400, 480, 500, 600, 750, 800, 1000, 1200, 1500, 2000, 2400, 3000, 4000, 6000, 12000, 150):
        bf = 48000 // baud
        tx = afskmodem.Transmitter(baud, 0.03)
        burst = tx.frames(bytes(rng.integers(0, 256, 3, dtype=np.uint8))) if baud == 12000 else \
            tx.wav_samples(bytes(rng.integers(0, 256, 3, dtype=np.uint8)))
        burst = burst[:-4700]
        pieces = []
        for i in range(400):
            L = int(rng.integers(4096, 9000))
            kind = i % 4
            if kind == 0:
                x = rng.integers(-32768, 32768, L).astype(np.int16)
            elif kind == 1:
                x = (rng.integers(-3000, 3000, L) * rng.integers(0, 12)).clip(-32768, 32767).astype(np.int16)
            else:
                x = rng.integers(-600, 600, L).astype(np.int16)
            if kind >= 2:
                at = int(rng.integers(0, max(1, L - len(burst))))
                seg = burst[: L - at]
                x[at: at + len(seg)] = seg
            pieces.append(x)
        ln = np.array([len(p) for p in pieces], np.int32)
        off = np.concatenate([[0], np.cumsum(ln[:-1], dtype=np.int64)]).astype(np.int64)
        flat = np.concatenate(pieces)
        bfa = np.full(len(pieces), bf, np.int32)
        for amp_end in (14000, 0, 22000):
            got = device_demod(torch, flat, off, ln, bfa, amp_end=amp_end, stride=64)
            want = O.demod_batch(flat, off, ln, bfa, amp_end, out_stride=64, n_threads=16)
            assert_same(got, want, f"fuzz baud {baud} amp_end {amp_end}")


def test_launch_is_graph_capture_safe(torch_cuda, entry):
    """The C-ABI launch path does no allocation / synchronisation, so a sequence of demod
    launches can be captured into a HIP graph and replayed (guideline: no hipMalloc / sync in
    the launch function).  Mixed entry: a three-rate batch with bit_frames on the device; uniform
    entry: one rate, bit_frames by value; grouped entry: the three-rate batch with a plan built BEFORE the
    capture (the first, uncaptured call builds and caches it; a plan launch is nothing but a kernel launch)."""
    torch = torch_cuda
    b = synth_batch(torch, 256, (2400,) if entry == "uniform" else (300, 1200, 2400), seed=99)
    if entry == "uniform":
        b["bf"] = 20
    if entry == "grouped":
        b["bf"] = b["h_bf"]                       # host array: nothing is read back from the device inside the capture
    stride = batch.out_stride_for(48000, 20)
    ref_out = batch.demod_batch(b["samples"], b["off"], b["ln"], b["bf"], 14000, out_stride=stride).cpu()
    outs = [batch.alloc_result(256, stride, "cuda:0") for _ in range(3)]
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            for o in outs:
                batch.demod_batch(b["samples"], b["off"], b["ln"], b["bf"], 14000, out=o, stream=side)
    for o in outs:
        o.flat.zero_()
    g.replay()
    torch.cuda.synchronize()
    for o in outs:
        got = o.cpu()
        for f in FIELDS:
            assert np.array_equal(getattr(got, f), getattr(ref_out, f)), f
        assert np.array_equal(got.bytes, ref_out.bytes)


def test_side_stream_launch_with_in_call_allocations(torch_cuda, entry):
    """demod_batch(stream=side) that allocates its result, its soft outputs and (mixed entry) the device
    copy of a host bit_frames list INSIDE the call: those fills / uploads run on torch's current stream
    and the launch on `side` must be ordered behind them.  The current stream is kept busy with a long
    fill so that a missing dependency would let the zero fill land after the kernel's stores.  Calls the
    real entry (not the per-rate splitter of the `entry` fixture, which scatters on the current stream)."""
    torch = torch_cuda
    bauds = (1200,) if entry == "uniform" else (300, 1200, 2400)
    b = synth_batch(torch, 192, bauds, seed=314)
    bf_h = b["bf"].cpu().numpy()
    stride = batch.out_stride_for(48000, 20)
    want = O.demod_batch(b["samples"].cpu().numpy(), np.arange(192, dtype=np.int64) * 48000,
                         np.full(192, 48000, np.int32), bf_h, 14000, out_stride=stride)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    busy = torch.empty(1 << 28, dtype=torch.int32, device="cuda:0")
    for rep in range(3):
        busy.fill_(rep)                          # ~1 GB of stores ahead of the in-call fills
        res = REAL_DEMOD_BATCH(b["samples"], b["off"], b["ln"], [int(v) for v in bf_h], 14000,
                               out_stride=stride, stream=side, diagnostics=True, margin_stride=48000 // 20,
                               entry=entry)
        side.synchronize()
        torch.cuda.synchronize()
        assert_same(res.cpu(), want, f"side stream rep {rep}")


def test_device_side_lengths_are_guarded_in_the_kernels(torch_cuda):
    """The device entries never see stream_len[] on the host: a NEGATIVE entry or one above
    AFSK_MAX_STREAM_LEN is refused by the kernel itself -- status AFSK_ST_BAD_LENGTH, empty record, no
    sample addressed -- inside a launch of 8256 streams (hint + warming armed) whose other streams decode
    bit-exactly as if the poisoned ones were not there.  Same for the gate (out_n_bursts = -1) and the
    modulator / noise generator (stream left untouched), whose bound is the caller's max_stream_len."""
    torch = torch_cuda
    dev = "cuda:0"
    n = 8256
    flat, off, ln, bf = large_launch_streams(n, (1200, 300, 2400, 375), 777)
    poison = {5: -1, 64: -2 ** 31, 4097: _native.MAX_STREAM_LEN + 1, 8191: 2 ** 31 - 1, 8255: -4096, 100: -48000}
    ln_p = ln.copy()
    for s_i, v in poison.items():
        ln_p[s_i] = v
    got = device_demod(torch, flat, off, ln_p, bf, stride=64)
    want = O.demod_batch(flat, off, ln, bf, 14000, out_stride=64, n_threads=os.cpu_count() or 16)
    bad = np.array(sorted(poison))
    good = np.setdiff1d(np.arange(n), bad)
    for f in FIELDS:
        assert np.array_equal(getattr(got, f)[good], want[f][good]), f
    m = np.arange(64)[None, :] < np.minimum(want["nbytes"][good], 64)[:, None]
    assert not ((got.bytes[good] != want["bytes"][good, :64]) & m).any()
    assert (got.status[bad] == _native.ST_BAD_LENGTH).all()
    assert (got.nbytes[bad] == 0).all() and (got.nbits[bad] == 0).all()
    assert (got.clock_idx[bad] == -1).all() and (got.term_frame[bad] == -1).all()
    # AFSK_MAX_STREAM_LEN itself is a legal length (test_maximum_stream_length decodes one); 0 is "too short"
    ln_z = ln.copy()
    ln_z[7] = 0
    assert device_demod(torch, flat, off, ln_z, bf, stride=64).status[7] == _native.ST_TOO_SHORT

    # gate: bound = max_stream_len of the call
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    caps = 64
    L = 20480
    rng = np.random.default_rng(12)
    x = (rng.integers(-30000, 30000, caps * L).astype(np.int16))
    g_ln = np.full(caps, L, np.int32)
    g_ln[[3, 17, 40]] = (-1, L + 1, 2 ** 31 - 1)
    g_off = np.arange(caps, dtype=np.int64) * L
    g = batch.gate_batch(t(x), t(g_off), t(g_ln), L, 18000, 14000, 4)
    torch.cuda.synchronize()
    nb = g.n_bursts.cpu().numpy()
    assert list(nb[[3, 17, 40]]) == [-1, -1, -1]
    for i in (0, 2, 4, 16, 18, 63):
        bursts, oe = O.gate_stream(x[i * L: (i + 1) * L], 18000, 14000, 4)
        assert int(nb[i]) == len(bursts) and int(g.open_end[i].item()) == oe
    owner, _, _ = g.burst_streams(t(g_off))
    assert not set(owner.cpu().numpy().tolist()) & {3, 17, 40}
    # (r6) the gate's own demodulator slots: a refused capture gets empty slots, every other one its bursts
    so, sl = g.slot_offset.cpu().numpy(), g.slot_len.cpu().numpy()
    assert not sl[[3, 17, 40]].any() and not so[[3, 17, 40]].any()
    for i in (0, 2, 4, 16, 18, 63):
        bursts, _ = O.gate_stream(x[i * L: (i + 1) * L], 18000, 14000, 4)
        assert [(int(so[i, k]) - i * L, int(sl[i, k])) for k in range(len(bursts))] == bursts
        assert not sl[i, len(bursts):].any()

    # modulator + noise: a poisoned stream keeps whatever the buffer held
    ns = 8
    m_ln = np.full(ns, 12000, np.int32)
    m_ln[[2, 5]] = (-7, 12001)
    m_off = np.arange(ns, dtype=np.int64) * 12000
    buf = torch.full((ns * 12000,), 1234, dtype=torch.int16, device=dev)
    payload = synth.payload_bytes(3, 0, ns, 4)
    batch.modulate_batch(t(payload), t(np.full(ns, 4, np.int32)), t(np.full(ns, 40, np.int32)),
                         t(np.full(ns, 30, np.int32)), t(m_off), t(m_ln), 12000, buf, True)
    batch.add_noise_batch(buf, t(m_off), t(m_ln), 12000, np.full(ns, synth.snr_to_scale_q24(20.0), np.int32), seed=1)
    torch.cuda.synchronize()
    h = buf.cpu().numpy().reshape(ns, 12000)
    assert (h[2] == 1234).all() and (h[5] == 1234).all()
    ok_ln = np.where(m_ln == 12000, 12000, 0).astype(np.int32)
    ref = O.modulate_batch(payload, np.full(ns, 4, np.int32), np.full(ns, 40, np.int32), np.full(ns, 30, np.int32),
                           m_off, ok_ln, ns * 12000, True).reshape(ns, 12000)
    for i in (0, 1, 3, 4, 6, 7):
        assert np.array_equal(h[i], O.add_noise(ref[i], 1, i, synth.snr_to_scale_q24(20.0))), i
