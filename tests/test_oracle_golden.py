"""Pin the CPU oracle (oracle/afsk_oracle.c) to vectors produced by the imported
reference (tests/golden/make_golden.py).  CPU-only."""
import hashlib

import numpy as np
import pytest

from oracle import afsk_oracle as O
from tests.golden_inputs import build_input, sha_i16


def test_templates(golden):
    for baud, t in golden["templates"].items():
        b = int(baud)
        assert O.space_tone(b).tolist() == t["space"]
        assert O.mark_tone(b).tolist() == t["mark"]
        assert O.training_cycle(b).tolist() == t["training"]


def test_baud_validity(golden):
    """ref:69-70 / :102-103 / :332 -- same accept/raise behaviour per baud."""
    for baud, e in golden["baud_validity"].items():
        b = int(baud)
        if e["construct"] != "ok":
            with pytest.raises(O.OracleError, match="Invalid baud rate"):
                O.space_tone(b), O.mark_tone(b)
            continue
        O.space_tone(b), O.mark_tone(b), O.training_cycle(b)
        frames = O.wav_convert(O.get_frames(b"Hi!", b))
        rt = e["roundtrip"]
        if rt == "ok":
            assert O.load_frames(frames, b) == b"Hi!"
        elif rt.startswith("bytes:"):
            assert O.load_frames(frames, b) == bytes.fromhex(rt[6:])
        elif "different lengths" in rt:
            with pytest.raises(O.OracleError, match="different lengths"):
                O.load_frames(frames, b)
        elif "IndexError" in rt:
            with pytest.raises(O.OracleError, match="list index out of range"):
                O.load_frames(frames, b)
        else:
            raise AssertionError(rt)


def test_ecc(golden):
    e = golden["ecc"]
    for k, v in e["codewords"].items():
        assert O.ecc_encode(k) == v
    for k, v in e["decode_table"].items():
        assert O.ecc_decode(k) == v
    for k, v in e["short"].items():
        assert O.ecc_encode(k) == v["encode"]
        assert O.ecc_decode(k) == v["decode"]


def test_frames_and_wav_quirk(golden):
    for c in golden["frames"]:
        fr = O.get_frames(bytes.fromhex(c["payload_hex"]), c["baud"], c["training_time"])
        assert len(fr) == c["n_frames"]
        assert sha_i16(fr) == c["frames_sha256"]
        wav = O.wav_convert(fr)
        assert len(wav) == c["n_wav"]
        assert sha_i16(wav) == c["wav_sha256"]
        assert int(np.count_nonzero(wav != fr[: len(wav)])) == c["wav_vs_ideal_diff"]


def test_primitives(golden):
    for p in golden["primitives"]:
        assert O.get_diff(p["a"], p["b"]) == p["diff"]
        assert O.get_amplitude(p["a"]) == p["amp_a"]
        assert O.amplify(p["a"]).tolist() == p["amplified_a"]


def test_decode_cases(golden):
    for c in golden["decode_cases"]:
        x = build_input(c)
        bits, ci, tf = O.decode_bits(x, c["baud"], c["amp_end"])
        assert ci == c["clock_idx"], c["tag"]
        assert tf == c["term_frame"], c["tag"]
        assert len(bits) == c["nbits"], c["tag"]
        assert hashlib.sha256(bits.encode()).hexdigest() == c["bits_sha256"], c["tag"]
        data = O.load_frames(x, c["baud"], c["amp_end"])
        assert data.hex() == c["bytes_hex"], c["tag"]


def test_demod_batch_matches_cases(golden):
    """The batched entry (the C-ABI's host twin) against the same cases, ragged batch."""
    cases = [c for c in golden["decode_cases"]]
    xs = [build_input(c) for c in cases]
    off = np.cumsum([0] + [len(x) for x in xs[:-1]]).astype(np.int64)
    ln = np.array([len(x) for x in xs], np.int32)
    bf = np.array([48000 // c["baud"] for c in cases], np.int32)
    flat = np.concatenate(xs)
    stride = max(160, max(c["nbytes"] for c in cases) + 8)        # the multi-second cases decode 100 - 900 bytes
    for amp_end in sorted({c["amp_end"] for c in cases}):
        out = O.demod_batch(flat, off, ln, bf, amp_end, out_stride=stride, n_threads=2)
        for i, c in enumerate(cases):
            if c["amp_end"] != amp_end:
                continue
            assert out["clock_idx"][i] == c["clock_idx"], c["tag"]
            assert out["term_frame"][i] == c["term_frame"], c["tag"]
            assert out["nbits"][i] == c["nbits"], c["tag"]
            assert out["nbytes"][i] == c["nbytes"], c["tag"]
            assert out["bytes"][i, : c["nbytes"]].tobytes().hex() == c["bytes_hex"], c["tag"]
            want = 1 if c["clock_idx"] == -1 else (2 if c["nbits"] == 0 else 0)
            assert out["status"][i] == want, c["tag"]


def test_soft_outputs_match_reference_internals(golden):
    """Row f4: the soft outputs are values the reference computes and discards -- the two
    getDiff results of __decodeBit (:346-347) and ECC.__decodeNibble's syndrome (:146-147) --
    recorded from the reference by make_golden.py for every decode case."""
    for c in golden["decode_cases"]:
        x = build_input(c)
        mstride = max(4000, c["soft"]["n_symbols"] + 8)
        out = O.demod_batch_soft(x, [0], [len(x)], [48000 // c["baud"]], c["amp_end"],
                                 out_stride=max(160, c["nbytes"] + 8), margin_stride=mstride)
        soft = c["soft"]
        ns = int(out["n_symbols"][0])
        assert ns == soft["n_symbols"], c["tag"]
        assert ns <= mstride
        m = out["margins"][0, :ns]
        assert m[:24].tolist() == soft["margins_head"], c["tag"]
        assert hashlib.sha256(m.astype("<i4").tobytes()).hexdigest() == soft["margins_sha256"], c["tag"]
        assert int(out["corrected"][0]) == soft["corrected"], c["tag"]
        assert out["bytes"][0, : c["nbytes"]].tobytes().hex() == c["bytes_hex"], c["tag"]
    assert sum(c["soft"]["corrected"] for c in golden["decode_cases"]) > 100   # exercised


def test_readme_roundtrip(golden):
    w = O.wav_convert(O.get_frames("Héellóo World!".encode(), 1200))
    assert O.load_frames(w, 1200).decode("utf-8") == golden["readme_roundtrip"]


def test_listen_gate_cases(golden):
    """Receiver.__listen (ref:299-319) replayed over captures: burst boundaries as recorded from
    the reference driven by a stub audio stream, and the decode of every closed burst."""
    from tests.golden_inputs import build_capture
    for c in golden["listen_cases"]:
        cap = build_capture(c["recipe"])
        assert len(cap) == c["n_samples"] and sha_i16(cap) == c["capture_sha256"], c["name"]
        bursts, open_end = O.gate_stream(cap, c["amp_start"], c["amp_end"], 16)
        assert open_end == c["open_end"], c["name"]
        assert bursts == [(b["start"], b["len"]) for b in c["bursts"]], c["name"]
        for (st, ln), b in zip(bursts, c["bursts"]):
            if ln == b["ref_len"]:      # closed burst: identical frames went into the reference
                assert O.load_frames(cap[st: st + ln], 1200, c["amp_end"]).hex() == b["bytes_hex"], c["name"]


def test_pure_python_restatement_matches_reference_vectors(golden):
    """oracle/pyref.py (the interpreter-speed twin timed by bench.py) on a spread of the
    reference-generated decode cases."""
    from oracle import pyref
    picked = [c for c in golden["decode_cases"]
              if c["tag"] in ("clean/hello/1200", "clean/A/300", "clean/fffefd/2400", "exact4096",
                              "too_short4000", "no_tail", "no_tail_plus1", "lead33", "amp_end_0",
                              "noise/1200/snr5/seed0", "noise/1200/snr0/seed2", "noise/2400/snr3")]
    assert len(picked) == 12
    for c in picked:
        x = build_input(c).tolist()
        data, nbits, ci, term = pyref.demod(x, 48000 // c["baud"], c["amp_end"])
        assert (ci, term, nbits, data.hex()) == (c["clock_idx"], c["term_frame"], c["nbits"],
                                                 c["bytes_hex"]), c["tag"]


def test_c_oracle_equals_pure_python_restatement_on_random_streams():
    """Two independent restatements of afskmodem.py:322-399 (scalar C, and line-by-line Python)
    agree on seeded random streams: loud noise, training fragments spliced at odd offsets,
    amplitudes around both squelch thresholds -- inputs none of the fixtures contain."""
    from oracle import pyref
    rng = np.random.default_rng(20261003)
    n = 0
    for trial in range(14):
        baud = (1200, 2400, 600, 300)[trial % 4]
        bf = 48000 // baud
        total = int(rng.integers(4096, 5200))
        x = rng.integers(-32768, 32768, total).astype(np.int16)
        if trial % 3 == 0:                                   # quieter noise: squelch decisions matter
            x = (x.astype(np.int32) * int(rng.integers(8, 20)) // 32).astype(np.int16)
        if trial % 2 == 0:                                   # a real burst somewhere inside
            w = O.wav_convert(O.get_frames(bytes(rng.integers(0, 256, 2, dtype=np.uint8)), baud, 0.02))
            w = w[: min(len(w), total - 7)]
            at = int(rng.integers(0, total - len(w)))
            x[at: at + len(w)] = w
        amp_end = int(rng.choice([0, 9000, 14000, 20000]))
        out = O.demod_batch(x, [0], [total], [bf], amp_end, out_stride=64, n_threads=1)
        data, nbits, ci, term = pyref.demod(x.tolist(), bf, amp_end)
        got = (int(out["clock_idx"][0]), int(out["term_frame"][0]), int(out["nbits"][0]),
               out["bytes"][0, : int(out["nbytes"][0])].tobytes())
        assert got == (ci, term, nbits, data), (trial, baud, total, amp_end)
        n += nbits > 0
    assert n >= 3                                            # some of them decode data
