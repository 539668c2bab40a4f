"""ctypes front end of oracle/libafsk_oracle.so -- TEST INFRASTRUCTURE ONLY.

The C file is a literal scalar restatement of /root/reference/afskmodem.py
(Waveforms :66-107, ECC :114-175, Receiver hot path :287-399, Transmitter frame
builder :452-469, wav writer quirk :239-244).  It is pinned against golden
vectors produced by the imported reference (tests/golden/make_golden.py).
Nothing in afskmodem_amd/ imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libafsk_oracle.so")

ERR_INVALID_BAUD = -1
ERR_LEN_MISMATCH = -2
ERR_CAPACITY = -3
ERR_EMPTY_SCAN = -4

ST_OK, ST_TOO_SHORT, ST_NO_DATA = 0, 1, 2


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (idempotent)."""
    src = os.path.join(_HERE, "afsk_oracle.c")
    hdr = os.path.join(_HERE, "afsk_oracle.h")
    stale = (not os.path.exists(_LIB_PATH)
             or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(src), os.path.getmtime(hdr)))
    if force or stale:
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "libafsk_oracle.so"])
    return _LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        i16p, u8p = C.POINTER(C.c_int16), C.POINTER(C.c_uint8)
        i32p, i64p = C.POINTER(C.c_int32), C.POINTER(C.c_int64)
        L.afsk_o_space_tone.argtypes = [C.c_int, i16p, C.c_int]
        L.afsk_o_mark_tone.argtypes = [C.c_int, i16p, C.c_int]
        L.afsk_o_training_cycle.argtypes = [C.c_int, i16p, C.c_int]
        L.afsk_o_get_amplitude.argtypes = [i16p, C.c_int]
        L.afsk_o_get_diff.argtypes = [i16p, i16p, C.c_int]
        L.afsk_o_amplify.argtypes = [i16p, C.c_int, i16p]
        L.afsk_o_amplify.restype = None
        L.afsk_o_recover_clock_index.argtypes = [i16p, C.c_int64, C.c_int, i16p, C.c_int]
        L.afsk_o_decode_bit.argtypes = [i16p, C.c_int, i16p, i16p]
        L.afsk_o_decode_bits.argtypes = [i16p, C.c_int64, C.c_int, C.c_int, u8p, C.c_int64,
                                         i32p, i64p]
        L.afsk_o_decode_bits.restype = C.c_int64
        for name in ("afsk_o_ecc_encode", "afsk_o_ecc_decode", "afsk_o_bits_to_bytes",
                     "afsk_o_bytes_to_bits"):
            f = getattr(L, name)
            f.argtypes = [u8p, C.c_int64, u8p]
            f.restype = C.c_int64
        L.afsk_o_frame_count.argtypes = [C.c_int, C.c_int, C.c_int64]
        L.afsk_o_frame_count.restype = C.c_int64
        L.afsk_o_get_frames.argtypes = [u8p, C.c_int64, C.c_int, C.c_int, i16p, C.c_int64]
        L.afsk_o_get_frames.restype = C.c_int64
        L.afsk_o_wav_convert.argtypes = [i16p, C.c_int64, i16p]
        L.afsk_o_wav_convert.restype = C.c_int64
        L.afsk_o_add_noise.argtypes = [i16p, C.c_int64, C.c_uint32, C.c_uint32, C.c_int32]
        L.afsk_o_add_noise.restype = None
        L.afsk_o_gate_stream.argtypes = [i16p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, i32p, i32p,
                                         i32p]
        L.afsk_o_gate_stream.restype = C.c_int32
        L.afsk_o_demod_batch.argtypes = [i16p, i64p, i32p, i32p, C.c_int32, C.c_int32, u8p,
                                         C.c_int32, i32p, i32p, i32p, i32p, i32p, C.c_int32]
        L.afsk_o_demod_stream_ex.argtypes = [i16p, C.c_int64, C.c_int, C.c_int, u8p, C.c_int32,
                                             i32p, i32p, i32p, i32p, i32p, i32p, i32p, C.c_int32]
        L.afsk_o_modulate_batch.argtypes = [u8p, C.c_int32, i32p, i32p, i32p, i64p, i32p,
                                            C.c_int32, C.c_int32, i16p]
        _lib = L
    return _lib


def _p(a: np.ndarray, ct):
    return a.ctypes.data_as(C.POINTER(ct))


def _i16(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.int16)


class OracleError(Exception):
    pass


_MESSAGES = {
    ERR_INVALID_BAUD: "Invalid baud rate.",
    ERR_LEN_MISMATCH: "Comparing two waveforms of different lengths.",
    ERR_CAPACITY: "oracle buffer capacity exceeded",
    ERR_EMPTY_SCAN: "list index out of range",
}


def _check(rc: int) -> int:
    if rc < 0:
        raise OracleError(_MESSAGES.get(int(rc), f"oracle error {rc}"))
    return int(rc)


def _template(fn, baud: int) -> np.ndarray:
    buf = np.zeros(4 * 48000, dtype=np.int16)
    n = _check(fn(int(baud), _p(buf, C.c_int16), buf.size))
    return buf[:n].copy()


def space_tone(baud: int) -> np.ndarray:
    return _template(lib().afsk_o_space_tone, baud)


def mark_tone(baud: int) -> np.ndarray:
    return _template(lib().afsk_o_mark_tone, baud)


def training_cycle(baud: int) -> np.ndarray:
    return _template(lib().afsk_o_training_cycle, baud)


def get_amplitude(frames) -> int:
    f = _i16(frames)
    return lib().afsk_o_get_amplitude(_p(f, C.c_int16), f.size)


def get_diff(a, b) -> int:
    a, b = _i16(a), _i16(b)
    if a.size != b.size:
        raise OracleError(_MESSAGES[ERR_LEN_MISMATCH])
    return lib().afsk_o_get_diff(_p(a, C.c_int16), _p(b, C.c_int16), a.size)


def amplify(chunk) -> np.ndarray:
    c = _i16(chunk)
    out = np.empty_like(c)
    lib().afsk_o_amplify(_p(c, C.c_int16), c.size, _p(out, C.c_int16))
    return out


def recover_clock_index(frames, baud: int) -> int:
    f = _i16(frames)
    tc = training_cycle(baud)
    rc = lib().afsk_o_recover_clock_index(_p(f, C.c_int16), f.size, int(48000 / baud),
                                          _p(tc, C.c_int16), tc.size)
    if rc < -1:
        _check(rc + 8)
    return rc


def decode_bits(frames, baud: int = 1200, amp_end_threshold: int = 14000):
    """-> (bit string, clock_idx, term_frame), as Receiver.__decodeBits (ref:354-381)."""
    f = _i16(frames)
    bits = np.zeros(f.size + 16, dtype=np.uint8)
    ci = C.c_int32(-1)
    term = C.c_int64(-1)
    n = _check(lib().afsk_o_decode_bits(_p(f, C.c_int16), f.size, int(baud),
                                        int(amp_end_threshold), _p(bits, C.c_uint8), bits.size,
                                        C.byref(ci), C.byref(term)))
    return "".join("1" if b else "0" for b in bits[:n]), ci.value, term.value


def _bits_arr(bits: str) -> np.ndarray:
    return np.frombuffer(bits.encode("ascii"), dtype=np.uint8) - ord("0")


def _bits_str(a: np.ndarray) -> str:
    return "".join("1" if b else "0" for b in a)


def ecc_encode(bits: str) -> str:
    b = np.ascontiguousarray(_bits_arr(bits))
    out = np.zeros(b.size * 2 + 8, dtype=np.uint8)
    n = lib().afsk_o_ecc_encode(_p(b, C.c_uint8), b.size, _p(out, C.c_uint8))
    return _bits_str(out[:n])


def ecc_decode(bits: str) -> str:
    b = np.ascontiguousarray(_bits_arr(bits))
    out = np.zeros(b.size + 8, dtype=np.uint8)
    n = lib().afsk_o_ecc_decode(_p(b, C.c_uint8), b.size, _p(out, C.c_uint8))
    return _bits_str(out[:n])


def bits_to_bytes(bits: str) -> bytes:
    b = np.ascontiguousarray(_bits_arr(bits))
    out = np.zeros(b.size // 8 + 1, dtype=np.uint8)
    n = lib().afsk_o_bits_to_bytes(_p(b, C.c_uint8), b.size, _p(out, C.c_uint8))
    return out[:n].tobytes()


def get_frames(data: bytes, baud: int = 1200, training_time: float = 0.5) -> np.ndarray:
    """Transmitter.__getFrames (ref:452-469); ts_cycles as ref:438."""
    ts_cycles = int(baud * training_time / 2)
    cap = _check(lib().afsk_o_frame_count(int(baud), ts_cycles, len(data)))
    out = np.zeros(cap, dtype=np.int16)
    d = np.frombuffer(bytes(data), dtype=np.uint8).copy() if len(data) else np.zeros(1, np.uint8)
    n = _check(lib().afsk_o_get_frames(_p(d, C.c_uint8), len(data), int(baud), ts_cycles,
                                       _p(out, C.c_int16), cap))
    return out[:n].copy()


def wav_convert(frames) -> np.ndarray:
    f = _i16(frames)
    out = np.zeros(f.size, dtype=np.int16)
    n = lib().afsk_o_wav_convert(_p(f, C.c_int16), f.size, _p(out, C.c_int16))
    return out[:n].copy()


def add_noise(samples, seed: int, stream_idx: int, scale_q24: int) -> np.ndarray:
    s = _i16(samples).copy()
    lib().afsk_o_add_noise(_p(s, C.c_int16), s.size, seed & 0xFFFFFFFF, stream_idx & 0xFFFFFFFF,
                           int(scale_q24))
    return s


def demod_batch(samples, stream_offset, stream_len, bit_frames, amp_end_threshold: int = 14000,
                out_stride: int = 128, n_threads: int = 1):
    """Host-pointer twin of afsk_demod_batch (include/afsk_amd.h). Returns a dict of arrays."""
    s = _i16(samples).reshape(-1)
    off = np.ascontiguousarray(stream_offset, dtype=np.int64)
    ln = np.ascontiguousarray(stream_len, dtype=np.int32)
    bf = np.ascontiguousarray(bit_frames, dtype=np.int32)
    n = off.size
    out = {
        "bytes": np.zeros((n, out_stride), dtype=np.uint8),
        "nbytes": np.zeros(n, dtype=np.int32),
        "nbits": np.zeros(n, dtype=np.int32),
        "clock_idx": np.zeros(n, dtype=np.int32),
        "term_frame": np.zeros(n, dtype=np.int32),
        "status": np.zeros(n, dtype=np.int32),
    }
    _check(lib().afsk_o_demod_batch(
        _p(s, C.c_int16), _p(off, C.c_int64), _p(ln, C.c_int32), _p(bf, C.c_int32),
        int(amp_end_threshold), n, _p(out["bytes"], C.c_uint8), out_stride,
        _p(out["nbytes"], C.c_int32), _p(out["nbits"], C.c_int32),
        _p(out["clock_idx"], C.c_int32), _p(out["term_frame"], C.c_int32),
        _p(out["status"], C.c_int32), int(n_threads)))
    return out


def demod_batch_soft(samples, stream_offset, stream_len, bit_frames,
                     amp_end_threshold: int = 14000, out_stride: int = 128,
                     margin_stride: int = 2400):
    """demod_batch plus the soft outputs of afsk_demod_batch_ex (corrected, margins);
    ``n_symbols`` = how many leading margins of each row the reference demodulated."""
    s = _i16(samples).reshape(-1)
    off = np.ascontiguousarray(stream_offset, dtype=np.int64)
    ln = np.ascontiguousarray(stream_len, dtype=np.int32)
    bf = np.ascontiguousarray(bit_frames, dtype=np.int32)
    n = off.size
    out = {k: np.zeros(n, dtype=np.int32)
           for k in ("nbytes", "nbits", "clock_idx", "term_frame", "status", "corrected")}
    out["bytes"] = np.zeros((n, out_stride), dtype=np.uint8)
    out["margins"] = np.zeros((n, margin_stride), dtype=np.int32)
    one = [np.zeros(1, dtype=np.int32) for _ in range(6)]
    for i in range(n):
        fr = np.ascontiguousarray(s[off[i]: off[i] + ln[i]])
        _check(lib().afsk_o_demod_stream_ex(
            _p(fr, C.c_int16), fr.size, int(bf[i]), int(amp_end_threshold),
            _p(out["bytes"][i], C.c_uint8), out_stride, *[_p(a, C.c_int32) for a in one],
            _p(out["margins"][i], C.c_int32), margin_stride))
        for k, a in zip(("nbytes", "nbits", "clock_idx", "term_frame", "status", "corrected"), one):
            out[k][i] = a[0]
    ci, tf = out["clock_idx"].astype(np.int64), out["term_frame"].astype(np.int64)
    out["n_symbols"] = np.where(ci >= 0, (tf - ci) // bf + out["nbits"], 0)
    return out


def modulate_batch(payload: np.ndarray, payload_len, bit_frames, ts_cycles, stream_offset,
                   stream_len, total_samples: int, wav_quirk: bool = True) -> np.ndarray:
    """Host-pointer twin of afsk_modulate_batch. payload is uint8 [n, stride]."""
    pl = np.ascontiguousarray(payload, dtype=np.uint8)
    n, stride = pl.shape
    plen = np.ascontiguousarray(payload_len, dtype=np.int32)
    bf = np.ascontiguousarray(bit_frames, dtype=np.int32)
    ts = np.ascontiguousarray(ts_cycles, dtype=np.int32)
    off = np.ascontiguousarray(stream_offset, dtype=np.int64)
    ln = np.ascontiguousarray(stream_len, dtype=np.int32)
    out = np.zeros(total_samples, dtype=np.int16)
    _check(lib().afsk_o_modulate_batch(_p(pl, C.c_uint8), stride, _p(plen, C.c_int32),
                                       _p(bf, C.c_int32), _p(ts, C.c_int32), _p(off, C.c_int64),
                                       _p(ln, C.c_int32), n, 1 if wav_quirk else 0,
                                       _p(out, C.c_int16)))
    return out


def gate_stream(frames, amp_start_threshold: int = 18000, amp_end_threshold: int = 14000,
                max_bursts: int = 16):
    """Receiver.__listen (ref:299-319) replayed over a capture -> ([(start, length)...], open_end)."""
    f = _i16(frames)
    st = np.zeros(max_bursts, np.int32)
    ln = np.zeros(max_bursts, np.int32)
    oe = C.c_int32(0)
    n = lib().afsk_o_gate_stream(_p(f, C.c_int16), f.size, int(amp_start_threshold),
                                 int(amp_end_threshold), int(max_bursts), _p(st, C.c_int32),
                                 _p(ln, C.c_int32), C.byref(oe))
    return [(int(st[i]), int(ln[i])) for i in range(n)], int(oe.value)


def load_frames(frames, baud: int = 1200, amp_end_threshold: int = 14000) -> bytes:
    """Receiver.load minus the file read and the utf-8 decode (ref:420-427)."""
    bits, _, _ = decode_bits(frames, baud, amp_end_threshold)
    if bits == "":
        return b""
    return bits_to_bytes(ecc_decode(bits))
