"""Differential fuzz of the native RIFF chunk walk (afsk_wav_probe: host-only, runs without a GPU)
against the stdlib `wave` reader -- the code SoundInput.loadFromFile calls in the reference
(afskmodem.py:213-217).  Contract under test (include/afsk_amd.h): whenever the native walk says
AFSK_WAV_OK, the byte range it reports is EXACTLY what readframes(getnframes()) returns; a file the
stdlib reader rejects must never be reported OK (the host then re-opens it with the stdlib reader so
that the caller sees the reference's own exception).  The reverse -- a readable file the native walk
declines -- only costs speed and is counted, not failed."""
from __future__ import annotations

import os
import struct
import wave

import numpy as np

from afskmodem_amd import _native, batch


def _chunk(cid: bytes, body: bytes, declared=None, pad=True) -> bytes:
    size = len(body) if declared is None else declared
    out = cid + struct.pack("<L", size & 0xFFFFFFFF) + body
    if pad and len(body) & 1:
        out += b"\x00"
    return out


def _fmt(rng) -> bytes:
    tag = int(rng.choice([1, 1, 1, 1, 3, 0xFFFE, 0, 7]))
    channels = int(rng.choice([1, 1, 2, 3, 0, 6]))
    bits = int(rng.choice([16, 16, 8, 24, 32, 12, 0, 4, 17]))
    rate = int(rng.choice([48000, 44100, 8000, 0, 1]))
    width = (bits + 7) // 8
    body = struct.pack("<HHLLHH", tag, channels, rate, rate * channels * width & 0xFFFFFFFF,
                       channels * width & 0xFFFF, bits)
    extra = int(rng.choice([0, 0, 0, 2, 3, 22]))
    body += bytes(rng.integers(0, 256, extra, dtype=np.uint8))
    if rng.random() < 0.08:
        body = body[: int(rng.integers(0, 16))]                  # a truncated fmt body
    return body


def _random_file(rng) -> bytes:
    parts = []
    n_pre = int(rng.choice([0, 0, 0, 1, 2]))
    order = ["fmt "] + ["junk"] * n_pre
    rng.shuffle(order)
    if rng.random() < 0.07:
        order = [o for o in order if o != "fmt "]                # no fmt chunk at all
    if rng.random() < 0.06:
        order = ["data0"] + order                                # a data chunk before fmt
    for o in order:
        if o == "fmt ":
            parts.append(_chunk(b"fmt ", _fmt(rng)))
        elif o == "data0":
            parts.append(_chunk(b"data", bytes(rng.integers(0, 256, int(rng.integers(0, 9)), dtype=np.uint8))))
        else:
            cid = bytes(rng.choice([b"LIST", b"junk", b"fact", b"cue ", b"JUNK", b"bext"]))
            n = int(rng.choice([0, 1, 2, 3, 4, 7, 26, 600]))
            parts.append(_chunk(cid, bytes(rng.integers(0, 256, n, dtype=np.uint8)),
                                pad=rng.random() > 0.05))
    nd = int(rng.choice([0, 1, 2, 3, 5, 8, 9, 64, 101, 1000, 4097]))
    body = bytes(rng.integers(0, 256, nd, dtype=np.uint8))
    declared = None
    r = rng.random()
    if r < 0.12:
        declared = nd + int(rng.choice([1, 2, 7, 100, 1 << 20]))   # claims more than the file holds
    elif r < 0.20:
        declared = max(0, nd - int(rng.choice([1, 2, 3])))         # claims less
    elif r < 0.23:
        declared = 0xFFFFFFFF
    parts.append(_chunk(b"data", body, declared, pad=rng.random() > 0.3))
    if rng.random() < 0.2:
        parts.append(_chunk(b"LIST", bytes(rng.integers(0, 256, int(rng.integers(0, 12)), dtype=np.uint8))))
    payload = b"".join(parts)
    riff = 4 + len(payload)
    r = rng.random()
    if r < 0.08:
        riff = max(0, riff - int(rng.choice([1, 2, 5, 9, 40])))     # RIFF size clips the form
    elif r < 0.14:
        riff += int(rng.choice([1, 2, 100, 1 << 24]))
    elif r < 0.16:
        riff = 0
    magic = b"RIFF" if rng.random() > 0.03 else bytes(rng.choice([b"RIFX", b"riff", b"FORM"]))
    form = b"WAVE" if rng.random() > 0.03 else bytes(rng.choice([b"AVI ", b"wave"]))
    blob = magic + struct.pack("<L", riff & 0xFFFFFFFF) + form + payload
    if rng.random() < 0.12:
        blob = blob[: int(rng.integers(0, len(blob) + 1))]          # truncated anywhere
    return blob


def _stdlib(fn: str):
    try:
        with wave.open(fn, "rb") as f:
            return f.readframes(f.getnframes())
    except BaseException:  # noqa: BLE001  (wave.Error, EOFError, struct.error ...: whatever the reference would see)
        return None


def test_native_riff_walk_differential_fuzz(tmp_path):
    rng = np.random.default_rng(20261006)
    n = 3000
    names, blobs = [], []
    for i in range(n):
        blob = _random_file(rng)
        fn = str(tmp_path / f"f{i}.wav")
        with open(fn, "wb") as f:
            f.write(blob)
        names.append(fn)
        blobs.append(blob)
    off, nbytes, status = batch.wav_probe(names)
    ok_both = declined = rejected = 0
    for fn, blob, o, nb, st in zip(names, blobs, off, nbytes, status):
        want = _stdlib(fn)
        if st == _native.WAV_OK:
            assert want is not None, (os.path.basename(fn), "native OK but the stdlib reader raises", blob[:64].hex())
            got = blob[int(o): int(o) + int(nb)]
            assert len(got) == int(nb), (os.path.basename(fn), "range outside the file")
            assert got == want, (os.path.basename(fn), len(got), len(want), blob[:64].hex())
            ok_both += 1
        elif want is None:
            rejected += 1
        else:
            declined += 1          # readable, but left to the stdlib fallback: legal, only slower
    # the generator must exercise all three outcomes, and the fast path must carry most readable files
    assert ok_both > 500 and rejected > 300, (ok_both, rejected, declined)
    assert declined <= ok_both // 10, (ok_both, rejected, declined)
