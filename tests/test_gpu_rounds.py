"""GPU parity (-m gpu) 4/5 -- the round geometry of the kernels: every alignment shift of the clock index, large
launches (L2 warming, tail hint), the clock-index-zero shortcut, the signal end and a dimmed symbol swept through every
position of a round, long outputs, run-time geometries in large launches.
(Split out of test_gpu_parity.py in r6; shared fixtures and helpers: tests/gpu_common.py.)"""
import os

import numpy as np
import pytest

import afskmodem_amd as afskmodem
from afskmodem_amd import _native, batch, synth
from oracle import afsk_oracle as O
from tests.golden_inputs import build_input
from tests.gpu_common import (FIELDS, LARGE_LAUNCH_BAUDS, REAL_DEMOD_BATCH, assert_same, device_demod, entry,  # noqa: F401
                              large_launch_streams, soft_demod, synth_batch, torch_cuda)

pytestmark = pytest.mark.gpu


def test_every_alignment_shift_and_short_tail_all_fast_bauds(torch_cuda):
    """The single-pass kernel re-aligns ring reads by (2*ci) & 15: exercise all 8 shifts, ring
    wrap-around on long streams, and tiny symbol counts, for every baud rate of the single-pass
    kernel (300 ... 12000 baud; for 800 / 500 / 480 / 400 baud also the mirror behind the ring
    that lets a lane piece run linearly past the ring end)."""
    torch = torch_cuda
    rng = np.random.default_rng(77)
    pieces, bfs = [], []
    for baud in (300, 600, 1200, 2400, 12000, 6000, 4000, 3000, 2000, 1500, 1000, 750, 800, 500, 480, 400):
        bf = 48000 // baud
        t = afskmodem.Transmitter(baud, 0.1)
        w_short = t.wav_samples(rng.integers(0, 256, 4, dtype=np.uint8).tobytes())
        w_long = afskmodem.Transmitter(baud, 0.3).wav_samples(
            rng.integers(0, 256, min(380, max(40, baud * 2 // 15)), dtype=np.uint8).tobytes())
        for lead in list(range(0, 9)) + [15, 16, 17, 511, 517, 1023, 2047, 3000]:
            pieces.append(np.concatenate([rng.integers(-400, 400, lead).astype(np.int16), w_short]))
            bfs.append(bf)
        for lead in (0, 3, 5, 12):
            pieces.append(np.concatenate([rng.integers(-400, 400, lead).astype(np.int16), w_long]))
            bfs.append(bf)
        # lengths hugging the 4096 window and the last-symbol rule (i < len - bf)
        base = np.concatenate([np.zeros(5, np.int16), afskmodem.Transmitter(baud, 0.0).wav_samples(b"ok")])
        base = np.concatenate([base, np.zeros(max(0, 4600 - len(base)), np.int16)])
        for L in (4096, 4097, 4096 + bf - 1, 4096 + bf, 4096 + bf + 1, 4096 + 2 * bf, 4500, len(base)):
            pieces.append(base[:L]); bfs.append(bf)
    ln = np.array([len(p) for p in pieces], np.int32)
    off = np.concatenate([[0], np.cumsum(ln[:-1], dtype=np.int64)]).astype(np.int64)
    bf = np.array(bfs, np.int32)
    flat = np.concatenate(pieces)
    stride = 400
    got = device_demod(torch, flat, off, ln, bf, stride=stride)
    want = O.demod_batch(flat, off, ln, bf, 14000, out_stride=stride, n_threads=8)
    assert_same(got, want, "alignment shifts")
    assert len(set(int(c) & 7 for c in got.clock_idx)) == 8      # all 8 shifts were exercised
    assert (got.nbytes > 100).any() and (got.nbits % 14 != 0).any()


@pytest.mark.parametrize("n", [6200, 8256])
def test_large_launch_arms_l2_warming_on_every_path(torch_cuda, entry, n):
    """Launches of 6144+ streams run the kernels with the tail hint (kHintMinStreams), from 8192 on also
    with the L2 warming requests behind the ring start (kWarmMinStreams, afsk_demod_ring.h); both shift
    the in-flight accounting of the rounds on every path.  6200 streams = hint only (the round loops
    switch to the dynamic wait_landed / fetch_through schedule without the warming requests in the
    count), 8256 = hint + warming.  Mixed entry: ONE launch cycling through all 16 compile-time rates
    plus a run-time-geometry rate; uniform entry: one launch of n streams PER RATE (every uniform
    kernel's large-launch form).  Every output equals the CPU oracle's."""
    import os
    torch = torch_cuda
    stride = 64
    threads = os.cpu_count() or 16
    if entry == "grouped":
        # five rates (fast and general-piece geometries), interleaved, in one rate-sorted launch of 2 n streams
        # (hint and warming armed), every wave reaching its stream through the index list
        flat, off, ln, bf = large_launch_streams(2 * n, (1200, 375, 300, 96, 6000), 4242)
        got = device_demod(torch, flat, off, ln, bf, stride=stride)
        want = O.demod_batch(flat, off, ln, bf, 14000, out_stride=stride, n_threads=threads)
        assert_same(got, want, "grouped large launch")
        assert (got.nbytes > 0).sum() > n // 2
    if entry in ("mixed", "grouped"):
        flat, off, ln, bf = large_launch_streams(n, LARGE_LAUNCH_BAUDS, 99)
        got = device_demod(torch, flat, off, ln, bf, stride=stride)
        want = O.demod_batch(flat, off, ln, bf, 14000, out_stride=stride, n_threads=threads)
        assert_same(got, want, "large launch")
        assert (got.nbytes > 0).sum() > n // 3
        # the same streams in a launch below the threshold give the same answers (hint / warming are timing only)
        sub = slice(0, 4096)
        got2 = device_demod(torch, flat[: int(off[4096])], off[sub], ln[sub], bf[sub], stride=stride)
        for f in FIELDS:
            assert np.array_equal(getattr(got2, f), getattr(got, f)[sub]), f
        return
    for baud in LARGE_LAUNCH_BAUDS + (150, 100, 375, 250, 240, 160, 120, 96, 80, 75, 48, 32, 24):
        flat, off, ln, bf = large_launch_streams(n, (baud,), 1000 + baud)
        dev = "cuda:0"
        res = REAL_DEMOD_BATCH(torch.from_numpy(flat).to(dev), torch.from_numpy(off).to(dev), torch.from_numpy(ln).to(dev),
                               48000 // baud, 14000, out_stride=stride, entry="uniform")
        torch.cuda.synchronize()
        want = O.demod_batch(flat, off, ln, bf, 14000, out_stride=stride, n_threads=threads)
        assert_same(res.cpu(), want, f"uniform large launch, {baud} baud")
        assert (want["nbytes"] > 0).sum() > (n // 4 if baud >= 300 else 0), baud


def test_clock_index_zero_without_a_search_boundaries(torch_cuda, entry):
    """r5: int(total(0) / N) == 0 is a mean no offset can undercut at the first index there is, so the kernels return
    clock index 0 without running the search (clock_index_is_zero).  The boundary: the first sample of a clean stream
    lowered by d gives total(0) = d exactly -- d = N - 1 still takes the shortcut, d = N does not (mean 1: the search must
    find whatever the reference finds, here an equal mean further on or offset 0 again); plus a stream whose copy of the
    training sequence starts one training period late (offset 0 is then NOT the minimum).  Every output equals the oracle's."""
    torch = torch_cuda
    streams, bfs = [], []
    for baud in (1200, 300, 600, 160, 800, 6000, 375):
        bf = 48000 // baud
        n2 = 2 * bf
        t = afskmodem.Transmitter(baud, 0.2)
        base = t.frames(b"ok!")                                   # ideal frames: total(0) == 0
        for d in (0, 1, n2 - 1, n2, n2 + 1, 2 * n2 - 1, 2 * n2, 3 * n2 + 5):
            x = base.copy()
            x[0] = np.int16(32767 - min(d, 65535))                # template is +32767 at sample 0: |32767 - x0| = d
            streams.append(x); bfs.append(bf)
        late = np.concatenate([np.zeros(n2, np.int16), base])     # the sequence starts one training period late
        streams.append(late); bfs.append(bf)
        noisy0 = base.copy()
        noisy0[: n2] = (noisy0[: n2].astype(np.int32) * 9 // 10).astype(np.int16)   # total(0) far above N, still the minimum region
        streams.append(noisy0); bfs.append(bf)
    ln = np.array([len(x) for x in streams], np.int32)
    off = np.concatenate([[0], np.cumsum(ln[:-1], dtype=np.int64)]).astype(np.int64)
    flat = np.concatenate(streams)
    bf = np.array(bfs, np.int32)
    got = device_demod(torch, flat, off, ln, bf, stride=16)
    want = O.demod_batch(flat, off, ln, bf, 14000, out_stride=16)
    assert_same(got, want, "clock index 0 shortcut boundaries")
    assert (want["clock_idx"] == 0).sum() >= 7 * 3 and (want["clock_idx"] > 0).sum() >= 7


def test_tail_hint_partial_rounds_with_the_signal_end_anywhere_in_a_round(torch_cuda, entry):
    """r5: for rounds of 6 KiB and more the tail-hint probes stand closer than a round, and the round that reaches past
    the hint is decoded from the symbols below the REQUESTED bytes first (a partial round); only if the squelch stop is
    not among them is the rest fetched and the round run again.  6200 one-second streams per rate whose payload length
    -- hence the position of the signal end inside its round -- sweeps from stream to stream, plus the cases that make
    the first guess wrong: a signal that ends in the middle of a symbol, noise in the silent tail, a second burst
    behind a gap (the hint holds back chunks that ARE needed), a level below the squelch threshold (every probe
    quiet), a stream cut right behind the data.  Every output equals the CPU oracle's, through all three entries."""
    import os
    torch = torch_cuda
    dev = "cuda:0"
    n, total = 6200, 48000
    threads = os.cpu_count() or 16
    rates = (4000, 800, 375, 96, 3000, 1200) if entry == "uniform" else (4000, 800, 375, 96, 3000, 1200, 500, 6000)
    rng = np.random.default_rng(515)

    def build(bauds):
        baud_a = np.asarray([bauds[i % len(bauds)] for i in range(n)], np.int32)
        bf = (48000 // baud_a).astype(np.int32)
        room = np.asarray([synth.one_second_payload(int(b)) for b in baud_a], np.int32)
        plen = np.maximum(room - (np.arange(n) // len(bauds)) % np.maximum(room, 1), 0).astype(np.int32)   # 0 ... room bytes
        payload = synth.payload_bytes(77, 0, n, int(room.max()))
        ts = np.asarray([synth.ts_cycles_for(int(b)) for b in baud_a], np.int32)
        off = np.arange(n, dtype=np.int64) * total
        ln = np.full(n, total, np.int32)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
        x = torch.zeros(n * total, dtype=torch.int16, device=dev)
        batch.modulate_batch(t(payload), t(plen), t(bf), t(ts), t(off), t(ln), total, x, True)
        h = x.cpu().numpy().reshape(n, total).copy()
        for i in range(n):
            k = i % 13
            if k == 3:                                        # the signal ends in the middle of a symbol
                e = int(np.flatnonzero(h[i])[-1]) if h[i].any() else 0
                h[i, max(e - int(rng.integers(1, 2 * bf[i])), 0):] = 0
            elif k == 5:                                      # noise in the tail, around the squelch threshold
                e = int(np.flatnonzero(h[i])[-1]) + 1 if h[i].any() else 0
                h[i, e:] = rng.integers(-22000, 22000, total - e)
            elif k == 7 and plen[i] * 14 * bf[i] < 12000:     # a second burst behind a gap: held-back chunks are needed
                e = int(np.flatnonzero(h[i])[-1]) + 1
                gap = int(rng.integers(2000, 9000))
                m = min(e, total - e - gap)
                if m > 4096:
                    h[i, e + gap: e + gap + m] = h[i, :m]
            elif k == 9:                                      # below the squelch threshold: every probe is quiet
                h[i] = (h[i].astype(np.int32) * 3 // 25).astype(np.int16)
            elif k == 11:                                     # the stream ends right behind the data
                e = int(np.flatnonzero(h[i])[-1]) + 1 if h[i].any() else total
                ln[i] = max(min(e + int(rng.integers(0, 3 * bf[i])), total), 4096)
        return h.reshape(-1), off, ln, bf

    if entry == "uniform":
        for baud in rates:
            flat, off, ln, bf = build((baud,))
            res = REAL_DEMOD_BATCH(torch.from_numpy(flat).to(dev), torch.from_numpy(off).to(dev), torch.from_numpy(ln).to(dev),
                                   48000 // baud, 14000, out_stride=64, entry="uniform")
            torch.cuda.synchronize()
            want = O.demod_batch(flat, off, ln, bf, 14000, out_stride=64, n_threads=threads)
            assert_same(res.cpu(), want, f"signal end sweep, {baud} baud")
            assert (want["nbytes"] > 0).sum() > n // 2, baud
    else:
        flat, off, ln, bf = build(rates)
        got = device_demod(torch, flat, off, ln, bf, stride=64)
        want = O.demod_batch(flat, off, ln, bf, 14000, out_stride=64, n_threads=threads)
        assert_same(got, want, "signal end sweep, eight rates in one launch")
        assert (want["nbytes"] > 0).sum() > n // 2


def test_long_outputs_with_the_tail_hint_armed(torch_cuda, entry):
    """Launches large enough to arm the tail hint (partial rounds that run twice: the receiver state goes back and the
    deferred Hamming flushes of the round are repeated) of streams that decode to 1.0 - 1.5 KiB each -- sixteen to
    twenty-four 64-byte flushes per stream, the bit buffer wrapping several times -- with payload lengths that move
    the signal end through the rounds and through the flush batches.  (r5; it also pinned the LDS output buffer that
    was measured and dropped: profiles/EXPERIMENTS.md, k22 / k23.)"""
    import os
    torch = torch_cuda
    dev = "cuda:0"
    threads = os.cpu_count() or 16
    n = 8256
    for baud, total in ((12000, 110000), (6000, 200000), (3000, 336000)):
        if entry == "uniform" and baud != 12000:
            continue                                  # (the uniform kernels of 6000 baud arm the hint from 16384 streams on)
        bf = 48000 // baud
        room = synth.one_second_payload(baud, stream_len=total)
        assert room > 1100, (baud, room)
        plen = (room - (np.arange(n) * 7) % 400).astype(np.int32)
        payload = synth.payload_bytes(123, 0, n, room)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
        off = np.arange(n, dtype=np.int64) * total
        ln = np.full(n, total, np.int32)
        x = torch.zeros(n * total, dtype=torch.int16, device=dev)
        batch.modulate_batch(t(payload), t(plen), t(np.full(n, bf, np.int32)), t(np.full(n, synth.ts_cycles_for(baud), np.int32)),
                             t(off), t(ln), total, x, False)
        stride = ((room + 63) // 64) * 64
        res = batch.demod_batch(x, t(off), t(ln), np.full(n, bf, np.int32), 14000, out_stride=stride)
        torch.cuda.synchronize()
        flat = x.cpu().numpy()
        want = O.demod_batch(flat, off, ln, np.full(n, bf, np.int32), 14000, out_stride=stride, n_threads=threads)
        assert_same(res.cpu(), want, f"long outputs, {baud} baud")
        assert (want["nbytes"] == plen).all(), baud
        del x, res


def test_squelch_stop_at_every_symbol_position_of_a_round(torch_cuda, entry):
    """r5: the squelch test of a data round / pass first asks one question per lane -- is the LARGEST quiet sum of my
    symbols still loud enough? (one zero test on the raw ballot where several lanes share a symbol) -- and only a round
    in which some lane says no forms the per-symbol amplitude words and locates the first quiet symbol.  Stream i of
    every rate has ONE symbol dimmed: symbol i - 8 counted from the first data symbol (the first eight lie in the
    training sequence and the terminator, where the reference does not look at the amplitude: ref:361-366), so the
    quiet symbol visits every lane and every piece of two 12000-baud rounds; a third of them are zeroed, a third
    scaled to just below the threshold, a third to just above it (no stop).  Rates: every phase-C family (ten / five /
    eight / four / two symbols per lane, one symbol per lane, two / four lanes per symbol, word-multiple and general
    pieces).  Every output equals the CPU oracle's."""
    import os
    torch = torch_cuda
    dev = "cuda:0"
    total = 48000
    threads = os.cpu_count() or 16
    rates = (12000, 6000, 4000, 3000, 1500, 2400, 1200, 600, 300, 500, 160, 96)

    def build(baud):
        bf = 48000 // baud
        room = synth.one_second_payload(baud)
        n = min(1400, 14 * room + 40)
        payload = synth.payload_bytes(91, 0, n, room)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
        off = np.arange(n, dtype=np.int64) * total
        ln = np.full(n, total, np.int32)
        x = torch.zeros(n * total, dtype=torch.int16, device=dev)
        batch.modulate_batch(t(payload), t(np.full(n, room, np.int32)), t(np.full(n, bf, np.int32)),
                             t(np.full(n, synth.ts_cycles_for(baud), np.int32)), t(off), t(ln), total, x, False)   # (no .wav quirk: it wipes out the 12000-baud mark tone)
        h = x.cpu().numpy().reshape(n, total).copy()
        clean = O.demod_batch(h[0], np.zeros(1, np.int64), ln[:1], np.full(1, bf, np.int32), 14000, out_stride=8)
        term = int(clean["term_frame"][0])
        assert term > 0 and clean["nbytes"][0] == room, (baud, term)
        for i in range(n):
            s0 = term + (i - 8) * bf
            if s0 < 0 or s0 + bf > total:
                continue
            sym = h[i, s0:s0 + bf].astype(np.float64)
            if i % 3 == 0:
                sym[:] = 0
            else:
                mean = np.abs(sym).mean()
                sym *= (14000.0 + (-0.6 if i % 3 == 1 else 0.6)) / max(mean, 1.0)
            h[i, s0:s0 + bf] = np.clip(np.rint(sym), -32768, 32767).astype(np.int16)
        return h.reshape(-1), off, ln, np.full(n, bf, np.int32)

    if entry == "uniform":
        for baud in rates:
            flat, off, ln, bf = build(baud)
            res = REAL_DEMOD_BATCH(torch.from_numpy(flat).to(dev), torch.from_numpy(off).to(dev), torch.from_numpy(ln).to(dev),
                                   48000 // baud, 14000, out_stride=64, entry="uniform")
            torch.cuda.synchronize()
            want = O.demod_batch(flat, off, ln, bf, 14000, out_stride=64, n_threads=threads)
            assert_same(res.cpu(), want, f"dimmed symbol sweep, {baud} baud")
            stopped = (want["nbits"] < 14 * synth.one_second_payload(baud)).sum()
            assert stopped >= min(len(ln) // 4, 8), (baud, stopped)     # the zeroed and the just-below symbols do stop the decode
    else:
        parts = [build(b) for b in rates]
        flat = np.concatenate([p[0] for p in parts])
        ln = np.concatenate([p[2] for p in parts])
        bf = np.concatenate([p[3] for p in parts])
        off = np.arange(len(ln), dtype=np.int64) * total
        perm = np.random.default_rng(4).permutation(len(ln))    # rates interleaved: the grouped entry sorts them back
        got = device_demod(torch, flat, off[perm], ln[perm], bf[perm], stride=64)
        want = O.demod_batch(flat, off[perm], ln[perm], bf[perm], 14000, out_stride=64, n_threads=threads)
        assert_same(got, want, "dimmed symbol sweep, twelve rates in one launch")


@pytest.mark.parametrize("n", [6200, 8256])
def test_uniform_runtime_geometry_large_launch(torch_cuda, entry, n):
    """The uniform kernel of the RUN-TIME geometry (bit_frames no Receiver can have -- not a divisor of
    48000 -- but the C-ABI accepts any multiple of 4): large launches arm its tail hint / L2 warming too.
    Streams come from the on-device modulator, which takes bit_frames directly."""
    if entry == "mixed":
        pytest.skip("uniform entry only (the mixed entry's large launches: test_large_launch_arms_l2_warming_on_every_path)")
    torch = torch_cuda
    dev = "cuda:0"
    for bf_v, total in ((136, 12000), (148, 13000), (1004, 44000)):
        plen = np.full(n, 2, np.int32)
        payload = synth.payload_bytes(bf_v, 0, n, 2)
        ts = np.full(n, max(2, 3000 // bf_v), np.int32)
        off = np.arange(n, dtype=np.int64) * total
        ln = np.full(n, total, np.int32)
        ln[::7] -= 4800 + (np.arange(0, n, 7) % 5)                       # some without the tail silence
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
        x = torch.zeros(n * total, dtype=torch.int16, device=dev)
        d_off, d_ln = t(off), t(ln)
        batch.modulate_batch(t(payload), t(plen), t(np.full(n, bf_v, np.int32)), t(ts), d_off, d_ln, total, x, False)
        q = np.where(np.arange(n) % 3 == 0, synth.snr_to_scale_q24(8.0), synth.snr_to_scale_q24(40.0)).astype(np.int32)
        batch.add_noise_batch(x, d_off, d_ln, total, q, seed=bf_v)
        stride = 16
        if entry == "grouped":
            res = REAL_DEMOD_BATCH(x, d_off, d_ln, np.full(n, bf_v, np.int32), 14000, out_stride=stride, validate=False,
                                   entry="grouped")
        else:
            res = REAL_DEMOD_BATCH(x, d_off, d_ln, bf_v, 14000, out_stride=stride, validate=False, entry="uniform")
        torch.cuda.synchronize()
        want = O.demod_batch(x.cpu().numpy(), off, ln, np.full(n, bf_v, np.int32), 14000, out_stride=stride,
                             n_threads=os.cpu_count() or 16)
        assert_same(res.cpu(), want, f"uniform run-time geometry, bit_frames {bf_v}, {n} streams")
        assert (want["nbytes"] == 2).sum() > n // 2, bf_v
