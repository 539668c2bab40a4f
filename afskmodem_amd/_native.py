"""ctypes binding of csrc/libafsk_amd.so (C-ABI: include/afsk_amd.h).

There is deliberately no fallback: if the HIP library is missing or no MI355X is
visible, every compute entry raises.  Nothing here (or anywhere in this package)
touches ``oracle/``.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# AFSK_AMD_LIB: another build of the same library (kernel A/B runs of bench.py with tools/build_variant.sh)
LIB_PATH = os.environ.get("AFSK_AMD_LIB") or os.path.join(_HERE, "csrc", "libafsk_amd.so")

OK = 0
E_INVALID_ARG, E_INVALID_BAUD, E_NO_DEVICE, E_HIP, E_HOST = -1, -2, -3, -4, -5
ST_OK, ST_TOO_SHORT, ST_NO_DATA, ST_INVALID_BAUD, ST_BAD_LENGTH = 0, 1, 2, 3, 4
WAV_OK = 0
WAV_SLOT = 5

SAMPLE_RATE = 48000
SYNC_WINDOW = 4096
MAX_STREAM_LEN = (1 << 30) - (1 << 15)      # AFSK_MAX_STREAM_LEN: 32-bit byte offsets in the kernels


class AfskNativeError(RuntimeError):
    """A C-ABI call returned a negative code."""

    def __init__(self, code: int, message: str):
        super().__init__(f"libafsk_amd error {code}: {message}")
        self.code = code


_lib = None

_i16p = C.POINTER(C.c_int16)
_u8p = C.POINTER(C.c_uint8)
_i32p = C.POINTER(C.c_int32)
_i64p = C.POINTER(C.c_int64)

# name -> (restype, argtypes); mirrors include/afsk_amd.h one to one
SIGNATURES = {
    "afsk_version": (C.c_int, []),
    "afsk_last_error": (C.c_int, [C.c_char_p, C.c_int]),
    "afsk_device_count": (C.c_int, []),
    "afsk_sync": (C.c_int, [C.c_void_p]),
    "afsk_demod_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                   C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "afsk_demod_batch_ex": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                      C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_int32, C.c_void_p]),
    "afsk_demod_batch_uniform": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                           C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_int32, C.c_void_p]),
    "afsk_group_plan_create": (C.c_int, [_i32p, C.c_int32, C.POINTER(C.c_void_p)]),
    "afsk_group_plan_create_ragged": (C.c_int, [_i32p, _i32p, C.c_int32, C.POINTER(C.c_void_p)]),
    "afsk_group_plan_info": (C.c_int, [C.c_void_p, _i32p, _i32p, _i32p, _i32p, C.c_int32]),
    "afsk_group_plan_destroy": (C.c_int, [C.c_void_p]),
    "afsk_demod_batch_grouped": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                           C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                           C.c_void_p]),
    "afsk_demod_batch_host": (C.c_int, [_i16p, C.c_int64, _i64p, _i32p, _i32p, C.c_int32,
                                        C.c_int32, _u8p, C.c_int32, _i32p, _i32p, _i32p, _i32p,
                                        _i32p]),
    "afsk_demod_streams_host": (C.c_int, [C.POINTER(C.c_void_p), _i32p, _i32p, C.c_int32, C.c_int32,
                                          _u8p, C.c_int32, _i32p, _i32p, _i32p, _i32p, _i32p]),
    "afsk_host_scratch_release": (C.c_int, []),
    "afsk_wav_probe": (C.c_int, [C.POINTER(C.c_char_p), C.c_int32, _i64p, _i64p, _i32p]),
    "afsk_file_sizes": (C.c_int, [C.POINTER(C.c_char_p), C.c_int32, _i64p]),
    "afsk_wav_ingest": (C.c_int, [C.POINTER(C.c_char_p), C.c_int32, _i64p, _i64p, C.c_void_p, C.c_int64,
                                  _i64p, _i64p, _i32p]),
    "afsk_wav_egress": (C.c_int, [C.POINTER(C.c_char_p), C.c_int32, C.c_void_p, _i64p, _i32p, _i32p]),
    "afsk_wav_upload": (C.c_int, [C.POINTER(C.c_char_p), _i64p, _i64p, _i64p, C.c_int32, C.c_void_p,
                                  C.c_int64]),
    "afsk_modulate_batch": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                      C.c_void_p, C.c_void_p]),
    "afsk_gate_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                  C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_void_p, C.c_void_p]),
    "afsk_gate_batch_slots": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                        C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "afsk_add_noise_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p,
                                       C.c_int32, C.c_uint32, C.c_uint32, C.c_void_p]),
}


def lib() -> C.CDLL:
    """Load the HIP shared library, failing loudly when it was not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with afskmodem_amd/csrc/build.sh "
                "(or __graft_entry__.build()); there is no CPU fallback")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def last_error() -> str:
    buf = C.create_string_buffer(1024)
    lib().afsk_last_error(buf, len(buf))
    return buf.value.decode("utf-8", "replace")


def check(rc: int) -> None:
    if rc != OK:
        raise AfskNativeError(rc, last_error())


def device_count() -> int:
    return int(lib().afsk_device_count())


def require_device() -> None:
    if device_count() <= 0:
        raise AfskNativeError(E_NO_DEVICE, "no HIP device visible: afskmodem_amd has no CPU fallback")
