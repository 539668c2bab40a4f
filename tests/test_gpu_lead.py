"""GPU parity (-m gpu), r6: streams with an ARBITRARY clock index at full size.

Every BASELINE config starts its training sequence at sample 0 (clock index 0, or a multiple of the training period
under noise).  A capture cut by the live gate (ref:299-319) does not: its burst starts anywhere inside a 2048-sample
block, so the clock index (ref:322-339) is any number and 7 of 8 streams have (2 * ci) & 15 != 0.  Since r6 the
kernel re-bases its LDS ring on the clock index (FastRing::rebase) instead of running a re-aligning form of every
round loop; these tests put 65536 such streams through every device entry: round trip on all of them, the CPU oracle
on 4096+, clock index == lead wherever the training tone survives the .wav writer intact."""
import numpy as np
import pytest

from afskmodem_amd import _native, batch, synth
from oracle import afsk_oracle as O

pytestmark = pytest.mark.gpu

FIELDS = ("nbytes", "nbits", "clock_idx", "term_frame", "status")
TOTAL = 48000


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert _native.device_count() > 0, "no HIP device: GPU tests need an MI355X"
    assert torch.cuda.is_available()
    return torch


def lead_batch(torch, n, bauds, seed, max_lead=2048, noise=600, wav_quirk=True):
    """n slots of 1 s: [lead_s samples of uniform noise, |x| < noise] [Transmitter frame, cut at the slot end]."""
    dev = "cuda:0"
    rng = np.random.default_rng(seed)
    baud = np.asarray([bauds[i % len(bauds)] for i in range(n)], np.int32)
    bf = (48000 // baud).astype(np.int32)
    plen = np.array([synth.one_second_payload(int(b)) for b in baud], np.int32)
    payload = synth.payload_bytes(seed, 0, n, int(plen.max()))
    ts = np.array([synth.ts_cycles_for(int(b)) for b in baud], np.int32)
    lead = rng.integers(0, max_lead, n).astype(np.int64)
    lead[: min(n, 16)] = np.arange(min(n, 16))                # every shift (2 * ci) & 15 at the very start, too
    off = np.arange(n, dtype=np.int64) * TOTAL
    ln = np.full(n, TOTAL, np.int32)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    x = torch.empty(n * TOTAL, dtype=torch.int16, device=dev)
    d_off, d_ln, d_bf = t(off), t(ln), t(bf)
    batch.modulate_batch(t(payload), t(plen), d_bf, t(ts), t(off + lead), t((ln - lead).astype(np.int32)), TOTAL, x, wav_quirk)
    x2 = x.view(n, TOTAL)
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    for s0 in range(0, n, 8192):
        s1 = min(n, s0 + 8192)
        nz = torch.randint(-(noise - 1), noise, (s1 - s0, max_lead), generator=gen, device=dev, dtype=torch.int16)
        j = torch.arange(max_lead, device=dev)[None, :]
        head = x2[s0:s1, :max_lead]
        head.copy_(torch.where(j < t(lead[s0:s1])[:, None], nz, head))
    torch.cuda.synchronize()
    return dict(x=x, off=d_off, ln=d_ln, bf=d_bf, h_off=off, h_ln=ln, h_bf=bf, payload=payload, plen=plen, lead=lead)


def check(torch, b, res, n_oracle, tag):
    got = res.cpu()
    n = len(b["h_bf"])
    pays = got.payloads()
    bad = [s for s in range(n) if pays[s] != b["payload"][s, : b["plen"][s]].tobytes()]
    assert not bad, f"{tag}: {len(bad)} of {n} streams do not decode to their payload, first {bad[:5]} (leads {b['lead'][bad[:5]]})"
    assert (got.status == 0).all()
    # oracle on an evenly spread sample (+ the first 64 streams: every shift)
    idx = np.unique(np.concatenate([np.arange(min(n, 64)), np.linspace(0, n - 1, n_oracle).astype(np.int64)]))
    xs = b["x"].view(n, TOTAL)[torch.from_numpy(idx).to(b["x"].device)].cpu().numpy().reshape(-1)
    off = np.arange(idx.size, dtype=np.int64) * TOTAL
    ln = np.full(idx.size, TOTAL, np.int32)
    stride = int(got.bytes.shape[1])
    want = O.demod_batch(xs, off, ln, b["h_bf"][idx], 14000, out_stride=stride, n_threads=16)
    for f in FIELDS:
        g, w = getattr(got, f)[idx], want[f]
        d = np.nonzero(g != w)[0]
        assert d.size == 0, f"{tag} {f}: {d.size} streams differ from the oracle, first {idx[d[:5]]}: got {g[d[:5]]} want {w[d[:5]]}"
    m = np.arange(stride)[None, :] < np.minimum(want["nbytes"], stride)[:, None]
    assert not ((got.bytes[idx] != want["bytes"][:, :stride]) & m).any(), f"{tag}: bytes differ from the oracle"
    return got


@pytest.mark.parametrize("entry", ["uniform", "mixed"])
def test_config5_with_random_lead_ins_full_size(torch_cuda, entry):
    """65536 x 1 s @1200 baud, every stream behind its own lead-in of 0 ... 2047 noise samples."""
    torch = torch_cuda
    n = 65536
    b = lead_batch(torch, n, (1200,), seed=6006)
    stride = batch.out_stride_for(TOTAL, 40)
    res = batch.demod_batch(b["x"], b["off"], b["ln"], 40 if entry == "uniform" else b["bf"], 14000,
                            out_stride=stride, entry=entry)
    torch.cuda.synchronize()
    got = check(torch, b, res, 4096, f"config5_lead/{entry}")
    # at 1200 baud the training cycle survives the .wav writer sample for sample: the first offset with mean 0 is the lead
    assert np.array_equal(got.clock_idx, b["lead"].astype(np.int32))
    assert (got.nbytes == 34).all() and (got.nbits == 476).all()
    shifts = (2 * got.clock_idx) & 15
    assert set(shifts.tolist()) == {0, 2, 4, 6, 8, 10, 12, 14}
    assert 0.85 < float((shifts != 0).mean()) < 0.90
    del b, res
    torch.cuda.empty_cache()


@pytest.mark.parametrize("entry", ["grouped", "mixed"])
def test_many_rates_with_random_lead_ins(torch_cuda, entry):
    """Every round-loop family behind random lead-ins in one large launch: fast (300 / 600 / 1200 / 2400), multi (12000 /
    6000 / 4000 / 3000 / 1500), watermark (800 / 500 / 400 / 375), general pieces (250 / 160 / 96 baud)."""
    torch = torch_cuda
    bauds = (300, 600, 1200, 2400, 6000, 4000, 3000, 1500, 800, 500, 400, 375, 250, 160, 96, 1000)
    n = 16384
    b = lead_batch(torch, n, bauds, seed=6007)
    stride = batch.out_stride_for(TOTAL, int(b["h_bf"].min()))
    res = batch.demod_batch(b["x"], b["off"], b["ln"], b["h_bf"] if entry == "grouped" else b["bf"], 14000,
                            out_stride=stride, entry=entry)
    torch.cuda.synchronize()
    got = check(torch, b, res, 4096, f"lead mix/{entry}")
    assert len(set(((2 * got.clock_idx) & 15).tolist())) == 8
    del b, res
    torch.cuda.empty_cache()


def test_uniform_kernels_with_random_lead_ins_small_and_large(torch_cuda):
    """Each uniform kernel in both of its forms (small launches: no hint; large: tail hint + L2 warming) on led-in
    streams, 12000 baud on ideal frames (the .wav writer destroys its mark tone, ref:239-244)."""
    torch = torch_cuda
    for baud, n in ((12000, 16384), (12000, 1024), (2400, 8192), (2400, 512), (300, 8192), (480, 4096), (480, 768),
                    (240, 4096), (1200, 1024), (120, 4096)):
        b = lead_batch(torch, n, (baud,), seed=6100 + baud + n, wav_quirk=baud != 12000)
        bf = 48000 // baud
        stride = batch.out_stride_for(TOTAL, bf)
        res = batch.demod_batch(b["x"], b["off"], b["ln"], bf, 14000, out_stride=stride, entry="uniform")
        torch.cuda.synchronize()
        check(torch, b, res, min(n, 512), f"{baud} baud x {n}")
        del b, res
    torch.cuda.empty_cache()


# ------------------------------------------------------------------ ragged batches (r6: length-aware plans)
def ragged_batch(torch, n, bauds, seed, lo=12000, hi=192000):
    """n streams back to back, lengths log-uniform in [lo, hi] samples (0.25 ... 4 s), each a whole Transmitter frame
    (training 0.1 s) with as many payload bytes as fit."""
    dev = "cuda:0"
    rng = np.random.default_rng(seed)
    baud = np.asarray([bauds[i % len(bauds)] for i in range(n)], np.int32)
    bf = (48000 // baud).astype(np.int32)
    ln = np.exp(rng.uniform(np.log(lo), np.log(hi), n)).astype(np.int32) & ~np.int32(7)
    ts = np.array([synth.ts_cycles_for(int(b), 0.1) for b in baud], np.int32)
    plen = np.maximum(0, (ln.astype(np.int64) - ts.astype(np.int64) * 2 * bf - 4 * bf - 4800) // (14 * bf.astype(np.int64))).astype(np.int32)
    payload = synth.payload_bytes(seed, 0, n, max(1, int(plen.max())))
    off = np.concatenate([[0], np.cumsum(ln[:-1], dtype=np.int64)]).astype(np.int64)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    x = torch.empty(int(ln.astype(np.int64).sum()), dtype=torch.int16, device=dev)
    d_off, d_ln, d_bf = t(off), t(ln), t(bf)
    batch.modulate_batch(t(payload), t(plen), d_bf, t(ts), d_off, d_ln, int(ln.max()), x, True)
    torch.cuda.synchronize()
    return dict(x=x, off=d_off, ln=d_ln, bf=d_bf, h_off=off, h_ln=ln, h_bf=bf, payload=payload, plen=plen)


def same_results(a, b, tag):
    for f in FIELDS:
        assert np.array_equal(getattr(a, f), getattr(b, f)), (tag, f)
    m = np.arange(a.bytes.shape[1])[None, :] < a.nbytes[:, None]
    assert not ((a.bytes != b.bytes) & m).any(), (tag, "bytes")


def test_ragged_lengths_longest_first_walk_changes_nothing_but_the_order(torch_cuda):
    """A ragged one-rate batch through the plain uniform launch (stream order), through a length-aware plan (the uniform
    kernel walking the longest-first list) and through demod_batch(stream_len_host=): identical outputs at the original
    stream numbers, equal to the payloads and -- on a sample -- to the CPU oracle.  Then four rates."""
    torch = torch_cuda
    n = 12288
    b = ragged_batch(torch, n, (1200,), seed=6200)
    stride = batch.out_stride_for(int(b["h_ln"].max()), 40)
    plain = batch.demod_batch(b["x"], b["off"], b["ln"], 40, 14000, out_stride=stride, entry="uniform")
    plan = batch.GroupPlan(b["h_bf"], "cuda:0", stream_len=b["h_ln"])
    assert plan.groups() == [(40, n)]
    walked = batch.demod_batch(b["x"], b["off"], b["ln"], None, 14000, out_stride=stride, plan=plan)
    auto = batch.demod_batch(b["x"], b["off"], b["ln"], 40, 14000, out_stride=stride, stream_len_host=b["h_ln"])
    torch.cuda.synchronize()
    p, w, a = plain.cpu(), walked.cpu(), auto.cpu()
    same_results(p, w, "plan vs stream order")
    same_results(p, a, "stream_len_host vs stream order")
    pays = w.payloads()
    assert all(pays[s] == b["payload"][s, : b["plen"][s]].tobytes() for s in range(n))
    idx = np.linspace(0, n - 1, 768).astype(np.int64)
    xs = b["x"].cpu().numpy()
    pieces = [xs[b["h_off"][i]: b["h_off"][i] + b["h_ln"][i]] for i in idx]
    ln = np.array([len(q) for q in pieces], np.int32)
    off = np.concatenate([[0], np.cumsum(ln[:-1], dtype=np.int64)]).astype(np.int64)
    want = O.demod_batch(np.concatenate(pieces), off, ln, b["h_bf"][idx], 14000, out_stride=stride, n_threads=16)
    for f in FIELDS:
        assert np.array_equal(getattr(w, f)[idx], want[f]), f
    del plain, walked, auto, plan, b
    # ---- four rates (the per-stream kernel walking rate buckets, each longest first) against stream order
    b = ragged_batch(torch, 8192, (1200, 300, 2400, 800), seed=6201)
    stride = batch.out_stride_for(int(b["h_ln"].max()), int(b["h_bf"].min()))
    mixed = batch.demod_batch(b["x"], b["off"], b["ln"], b["bf"], 14000, out_stride=stride, entry="mixed")
    walked = batch.demod_batch(b["x"], b["off"], b["ln"], b["h_bf"], 14000, out_stride=stride, stream_len_host=b["h_ln"])
    torch.cuda.synchronize()
    same_results(mixed.cpu(), walked.cpu(), "four rates, ragged")
    pays = walked.cpu().payloads()
    assert all(pays[s] == b["payload"][s, : b["plen"][s]].tobytes() for s in range(8192))
    del mixed, walked, b
    torch.cuda.empty_cache()
