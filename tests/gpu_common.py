"""Shared by the GPU parity files (tests/test_gpu_*.py): the device fixture, the `entry` fixture that runs every test
through the three device entries, comparison and batch-building helpers.  Not a test module.

GPU parity tests (-m gpu) compare the HIP path, called through the C-ABI, against
 (a) the committed golden vectors produced by the reference itself, and
 (b) the CPU oracle on the same seeded inputs,
bit-exact in every output (bytes, nbytes, nbits, clock index, terminator frame,
status).  Integer path: tolerance is zero."""
import os

import numpy as np
import pytest

import afskmodem_amd as afskmodem
from afskmodem_amd import _native, batch, synth
from oracle import afsk_oracle as O
from tests.golden_inputs import build_input


FIELDS = ("nbytes", "nbits", "clock_idx", "term_frame", "status")


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    # fail loudly: GPU tests must exercise the HIP library, never a fallback
    assert _native.device_count() > 0, "no HIP device: GPU tests need an MI355X"
    assert torch.cuda.is_available()
    return torch


REAL_DEMOD_BATCH = batch.demod_batch


@pytest.fixture(autouse=True, params=["mixed", "uniform", "grouped"])
def entry(request, monkeypatch):
    """Every test of this file runs three times: with every device launch forced through the
    per-stream entry (afsk_demod_batch / _ex: the mixed-baud kernel), through the
    Receiver-shaped afsk_demod_batch_uniform (one kernel per bit_frames), and through the rate-grouped
    dispatch afsk_demod_batch_grouped (a plan built from the host-side rates: ONE launch of the per-stream
    kernel that walks the streams bucket by bucket through an index list once four or more rates are mixed,
    in stream order below that, the uniform kernel for one rate; outputs at the original stream numbers).  A batch with several baud rates is, in the second run, split BY THE TEST into one
    uniform launch per rate and scattered back into one result -- so every parity case below pins both
    kernel families and both ways of reaching the second.  (Host entries pick the uniform kernel or the
    grouped dispatch themselves, from their bit_frames array.)"""
    mode = request.param

    def wrapped(samples, stream_offset, stream_len, bit_frames, amp_end_threshold=14000, out=None,
                out_stride=None, stream=None, validate=True, diagnostics=False, margin_stride=None, entry="auto",
                plan=None, stream_len_host=None):
        import torch
        n = int(stream_offset.numel())
        kw = dict(stream=stream, diagnostics=diagnostics, margin_stride=margin_stride)
        if mode == "mixed":
            return REAL_DEMOD_BATCH(samples, stream_offset, stream_len, bit_frames, amp_end_threshold, out=out,
                                    out_stride=out_stride, validate=validate, entry="mixed", **kw)
        bf_h = (bit_frames.cpu().numpy() if isinstance(bit_frames, torch.Tensor)
                else np.broadcast_to(np.asarray(bit_frames, np.int32), (n,)))
        if validate and n:
            batch.validate_bit_frames(bf_h)
        if mode == "grouped":
            # (host-side lengths, when the caller has them -- load_batch, decode_captures --, go on to the plan: a ragged
            # batch is then walked longest first)
            return REAL_DEMOD_BATCH(samples, stream_offset, stream_len, np.ascontiguousarray(bf_h), amp_end_threshold,
                                    out=out, out_stride=out_stride, validate=False, entry="grouped",
                                    stream_len_host=stream_len_host, **kw)
        values = sorted(set(int(v) for v in bf_h))
        if len(values) <= 1:
            return REAL_DEMOD_BATCH(samples, stream_offset, stream_len, values[0] if values else 40,
                                    amp_end_threshold, out=out, out_stride=out_stride, validate=False,
                                    entry="uniform", **kw)
        if out is None:
            out = batch.alloc_result(n, int(out_stride), samples.device)
        stride = int(out.bytes.shape[1])
        if diagnostics:
            if out.corrected is None:
                out.corrected = torch.zeros(n, dtype=torch.int32, device=samples.device)
            if out.margins is None:
                out.margins = torch.zeros((n, int(margin_stride)), dtype=torch.int32, device=samples.device)
            kw["margin_stride"] = int(out.margins.shape[1])
        for v in values:
            idx = torch.from_numpy(np.nonzero(bf_h == v)[0]).to(samples.device)
            part = REAL_DEMOD_BATCH(samples, stream_offset[idx].contiguous(), stream_len[idx].contiguous(), v,
                                    amp_end_threshold, out_stride=stride, validate=False, entry="uniform", **kw)
            for f in ("bytes", "nbytes", "nbits", "clock_idx", "term_frame", "status") + (("corrected", "margins") if diagnostics else ()):
                getattr(out, f)[idx] = getattr(part, f)
        return out

    monkeypatch.setattr(batch, "demod_batch", wrapped)
    return mode


def assert_same(got, want, tag=""):
    """got: HostDemodResult, want: oracle dict."""
    for f in FIELDS:
        g, w = getattr(got, f), want[f]
        bad = np.nonzero(g != w)[0]
        assert bad.size == 0, f"{tag} {f}: {bad.size} streams differ, first {bad[:5]}: got {g[bad[:5]]} want {w[bad[:5]]}"
    stride = min(got.bytes.shape[1], want["bytes"].shape[1])
    n = np.minimum(want["nbytes"], stride)
    col = np.arange(stride)[None, :]
    mask = col < n[:, None]
    diff = (got.bytes[:, :stride] != want["bytes"][:, :stride]) & mask
    bad = np.nonzero(diff.any(axis=1))[0]
    assert bad.size == 0, f"{tag} bytes: {bad.size} streams differ, first {bad[:5]}"


def device_demod(torch, flat, off, ln, bf, amp_end=14000, stride=None):
    dev = "cuda:0"
    x = torch.from_numpy(np.ascontiguousarray(flat, dtype=np.int16)).to(dev)
    o = torch.from_numpy(np.ascontiguousarray(off, dtype=np.int64)).to(dev)
    l = torch.from_numpy(np.ascontiguousarray(ln, dtype=np.int32)).to(dev)
    if stride is None:
        stride = batch.out_stride_for(int(np.max(ln)), int(np.min(bf)))
    res = batch.demod_batch(x, o, l, np.asarray(bf, np.int32), amp_end, out_stride=stride)
    torch.cuda.synchronize()
    return res.cpu()


def soft_demod(torch, flat, off, ln, bf, amp_end, stride, mstride):
    dev = "cuda:0"
    x = torch.from_numpy(np.ascontiguousarray(flat, dtype=np.int16)).to(dev)
    o = torch.from_numpy(np.ascontiguousarray(off, dtype=np.int64)).to(dev)
    l = torch.from_numpy(np.ascontiguousarray(ln, dtype=np.int32)).to(dev)
    res = batch.demod_batch(x, o, l, np.asarray(bf, np.int32), amp_end, out_stride=stride,
                            diagnostics=True, margin_stride=mstride)
    torch.cuda.synchronize()
    nsym = res.symbols_demodulated(np.asarray(bf, np.int64)).cpu().numpy()
    return res.cpu(), res.corrected.cpu().numpy(), res.margins.cpu().numpy(), nsym


def synth_batch(torch, n, bauds, seed, total=48000, training_time=0.5, snr_db=None,
                payload_len=None, wav_quirk=True):
    """Modulate n streams on the GPU; returns device tensors + host copies."""
    dev = "cuda:0"
    bauds = np.asarray([bauds[i % len(bauds)] for i in range(n)], np.int32)
    bf = (48000 // bauds).astype(np.int32)
    plen = np.array([payload_len if payload_len is not None else synth.one_second_payload(int(b))
                     for b in bauds], np.int32)
    stride = int(plen.max()) if n else 1
    payload = synth.payload_bytes(seed, 0, n, max(stride, 1))
    ts = np.array([synth.ts_cycles_for(int(b), training_time) for b in bauds], np.int32)
    off = (np.arange(n, dtype=np.int64) * total)
    ln = np.full(n, total, np.int32)
    t = lambda a: torch.from_numpy(a).to(dev)  # noqa: E731
    samples = torch.empty(n * total, dtype=torch.int16, device=dev)
    d_off, d_ln, d_bf = t(off), t(ln), t(bf)
    batch.modulate_batch(t(payload), t(plen), d_bf, t(ts), d_off, d_ln, total, samples, wav_quirk)
    if snr_db is not None:
        q = np.asarray([synth.snr_to_scale_q24(s) for s in np.broadcast_to(snr_db, (n,))], np.int32)
        batch.add_noise_batch(samples, d_off, d_ln, total, q, seed=seed + 1, stream_idx_base=0)
    else:
        q = None
    torch.cuda.synchronize()
    return dict(samples=samples, off=d_off, ln=d_ln, bf=d_bf, h_off=off, h_ln=ln, h_bf=bf,
                payload=payload, plen=plen, ts=ts, q=q, total=total)


LARGE_LAUNCH_BAUDS = (300, 400, 480, 500, 600, 750, 800, 1000, 1200, 1500, 2000, 2400, 3000, 4000, 6000, 12000, 200)


def large_launch_streams(n, bauds, seed):
    """n short ragged streams cycling through `bauds`: lengths around the 24 KiB the L2 warming covers,
    odd leads, some noisy, and streams that defeat the tail hint of large launches (sparse amplitude
    probes decide how far to prefetch): a signal weaker than the squelch threshold (every probe
    "quiet", yet the training phase decodes it), a late start behind silence, two bursts with a gap."""
    rng = np.random.default_rng(seed)
    protos = {}
    for baud in bauds:
        t = afskmodem.Transmitter(baud, 0.08)
        ws = []
        for k in range(4):
            data = rng.integers(0, 256, 3 + k, dtype=np.uint8).tobytes()
            ws.append(t.frames(data) if baud == 12000 else t.wav_samples(data))
        protos[baud] = ws
    pieces, bfs = [], []
    for i in range(n):
        baud = bauds[i % len(bauds)]
        w = protos[baud][(i // len(bauds)) % 4]
        lead = int(rng.integers(0, 40)) if i % 3 else 0
        x = np.concatenate([np.zeros(lead, np.int16), w])
        L = int(rng.integers(11000, 15000)) if i % 5 else len(x)
        x = x[:L] if L <= len(x) else np.concatenate([x, np.zeros(L - len(x), np.int16)])
        if i % 7 == 0:
            x = np.clip(x.astype(np.int32) + rng.integers(-6000, 6000, len(x)), -32768, 32767).astype(np.int16)
        if i % 11 == 3:
            x = (x.astype(np.int32) * 3 // 25).astype(np.int16)
        elif i % 11 == 5:
            x = np.concatenate([np.zeros(int(rng.integers(3000, 9000)), np.int16), x])
        elif i % 11 == 8:
            x = np.concatenate([x[: len(x) // 2], np.zeros(int(rng.integers(1500, 5000)), np.int16), x])
        pieces.append(x); bfs.append(48000 // baud)
    ln = np.array([len(p) for p in pieces], np.int32)
    off = np.concatenate([[0], np.cumsum(ln[:-1], dtype=np.int64)]).astype(np.int64)
    return np.concatenate(pieces), off, ln, np.array(bfs, np.int32)
