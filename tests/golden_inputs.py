"""Rebuild the sample arrays of tests/golden/reference_vectors.json from their
recorded generation parameters (test helper; uses the CPU oracle's framing and
noise generator, both themselves pinned by SHA-256 values in the fixture)."""
from __future__ import annotations

import hashlib

import numpy as np

from oracle import afsk_oracle as O


def sha_i16(a) -> str:
    return hashlib.sha256(np.asarray(a, dtype="<i2").tobytes()).hexdigest()


def _wav(payload_hex: str, baud: int, training_time: float, total=None) -> np.ndarray:
    w = O.wav_convert(O.get_frames(bytes.fromhex(payload_hex), baud, training_time))
    if total is not None:
        w = np.concatenate([w, np.zeros(total - len(w), np.int16)])
    return w


def build_input(case: dict) -> np.ndarray:
    g = case["gen"]
    kind = g["kind"]
    if kind == "wav":
        x = _wav(g["payload_hex"], g["baud"], g["training_time"], g.get("total"))
    elif kind == "zeros":
        x = np.zeros(g["total"], np.int16)
    elif kind == "wav_trunc":
        x = _wav(g["payload_hex"], g["baud"], g["training_time"])[: g["trunc"]]
    elif kind == "training_only":
        x = np.tile(O.training_cycle(g["baud"]), g["cycles"])
    elif kind == "frames_trunc_tail":
        fr = O.get_frames(bytes.fromhex(g["payload_hex"]), g["baud"], g["training_time"])[:-4800]
        x = np.concatenate([fr, np.zeros(g["extra"], np.int16)])
    elif kind == "wav_lead":
        x = np.concatenate([np.zeros(g["lead"], np.int16),
                            _wav(g["payload_hex"], g["baud"], g["training_time"])])
    elif kind == "wav_noise":
        x = _wav(g["payload_hex"], g["baud"], g["training_time"], g["total"])
        x = O.add_noise(x, g["seed"], g["stream_idx"], g["scale_q24"])
    elif kind == "garbage":
        x = O.add_noise(np.zeros(g["total"], np.int16), g["seed"], g["stream_idx"], g["scale_q24"])
    elif kind == "recipe":
        x = build_capture(g["recipe"])
        if g.get("noise"):
            seed, stream_idx, scale_q24 = g["noise"]
            x = O.add_noise(x, seed, stream_idx, scale_q24)
    else:
        raise ValueError(kind)
    x = np.ascontiguousarray(x, dtype=np.int16)
    assert len(x) == case["n_samples"], (case["tag"], len(x), case["n_samples"])
    assert sha_i16(x) == case["input_sha256"], case["tag"]
    return x


def build_capture(recipe) -> np.ndarray:
    """A long capture from a list of segments (used by the live-gate cases):
    ["zeros", n] | ["noise", n, seed, scale_q24] (oracle integer noise on silence) |
    ["burst", payload_hex, baud, training_time, keep] (wav samples, first `keep` if not None) |
    ["frames", payload_hex, baud, training_time, keep] (the ideal Transmitter frames: 12000 baud, whose mark tone
    the wav writer's decimate / duplicate quirk destroys) |
    ["square", n, amplitude] (+a, -a, +a, ...) |
    ["burst_level", payload_hex, baud, training_time, keep, level] (the burst with +level / -level
    instead of full scale; silence stays 0) |
    ["burst_dc", payload_hex, baud, training_time, keep, num, den, dc] (burst * num // den + dc, clipped
    to int16)."""
    parts = []
    for seg in recipe:
        kind = seg[0]
        if kind == "zeros":
            parts.append(np.zeros(seg[1], np.int16))
        elif kind == "noise":
            parts.append(O.add_noise(np.zeros(seg[1], np.int16), seg[2], 0, seg[3]))
        elif kind == "burst":
            w = _wav(seg[1], seg[2], seg[3])
            parts.append(w if seg[4] is None else w[: seg[4]])
        elif kind == "frames":                       # the IDEAL Transmitter frames (no wav-writer quirk)
            w = O.get_frames(bytes.fromhex(seg[1]), seg[2], seg[3])
            parts.append(w if seg[4] is None else w[: seg[4]])
        elif kind == "square":
            parts.append(np.tile(np.array([seg[2], -seg[2]], np.int16), seg[1] // 2))
        elif kind == "burst_level":
            w = _wav(seg[1], seg[2], seg[3])
            w = w if seg[4] is None else w[: seg[4]]
            parts.append((np.sign(w.astype(np.int32)) * int(seg[5])).astype(np.int16))
        elif kind == "burst_dc":
            w = _wav(seg[1], seg[2], seg[3])
            w = (w if seg[4] is None else w[: seg[4]]).astype(np.int64)
            parts.append(np.clip(w * int(seg[5]) // int(seg[6]) + int(seg[7]), -32768, 32767).astype(np.int16))
        else:
            raise ValueError(kind)
    return np.ascontiguousarray(np.concatenate(parts), dtype=np.int16)


# ------------------------------------------------------------------ raw RIFF files (wav ingest)


def wav_body(n: int) -> bytes:
    """The byte pattern every wav_ingest case fills its data chunk with."""
    return bytes(((i * 37 + 11) ^ (i >> 3)) & 0xFF for i in range(n))


def fmt_body(tag=1, channels=1, rate=48000, bits=16, extra=b"") -> bytes:
    import struct
    width = (bits + 7) // 8
    return struct.pack("<HHLLHH", tag, channels, rate, rate * channels * width, channels * width, bits) + extra


def build_riff(recipe: dict) -> bytes:
    """File bytes of a wav_raw_cases recipe: {"magic", "form", "riff_size" (int or "auto"),
    "chunks": [[id, kind, arg, declared_size or None], ...]}; kind "hex" -> bytes.fromhex(arg),
    "pattern" -> wav_body(arg), "fmt" -> fmt_body(**arg).  A chunk body is padded to an even
    length (RIFF rule) unless its entry carries a fifth element False."""
    import struct
    out = b""
    for ent in recipe["chunks"]:
        cid, kind, arg, declared = ent[0], ent[1], ent[2], ent[3]
        pad = ent[4] if len(ent) > 4 else True
        if kind == "fmt":
            kw = dict(arg)
            if "extra_hex" in kw:
                kw["extra"] = bytes.fromhex(kw.pop("extra_hex"))
            body = fmt_body(**kw)
        else:
            body = bytes.fromhex(arg) if kind == "hex" else wav_body(arg)
        size = len(body) if declared is None else declared
        out += cid.encode("latin-1") + struct.pack("<L", size) + body
        if pad and len(body) & 1:
            out += b"\x00"
    riff = recipe["riff_size"]
    if riff == "auto":
        riff = 4 + len(out)
    return recipe["magic"].encode("latin-1") + struct.pack("<L", riff) + recipe["form"].encode("latin-1") + out
