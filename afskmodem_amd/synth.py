"""Synthetic workload helpers (host side): payloads, layouts, noise scale.

These build the inputs of BASELINE.json's configs -- Transmitter-generated
streams of a fixed length -- without shipping audio: payload bytes come from a
counter hash (stable across numpy versions), the waveform is produced on the GPU
by ``batch.modulate_batch`` (or on the host by ``Transmitter.wav_samples``).
"""
from __future__ import annotations

import numpy as np

SAMPLE_RATE = 48000
TAIL_SILENCE = 4800

# payload bytes that make an exactly-1-s stream at training_time 0.5
ONE_SECOND_PAYLOAD = {300: 8, 600: 16, 1200: 34, 2400: 68}


def one_second_payload(baud: int, training_time: float = 0.5, stream_len: int = SAMPLE_RATE) -> int:
    """Largest payload (bytes) whose Transmitter frames (ref:452-469: training, terminator, 14
    symbols per byte, 4800 tail zeros) fit stream_len samples; equals ONE_SECOND_PAYLOAD for the
    BASELINE bauds."""
    bf = SAMPLE_RATE // int(baud)
    room = stream_len - ts_cycles_for(baud, training_time) * 2 * bf - 4 * bf - TAIL_SILENCE
    return max(room // (14 * bf), 0)


def _hash32(x: np.ndarray) -> np.ndarray:
    x = x.astype(np.uint32)
    x ^= x >> np.uint32(16)
    x = (x * np.uint32(0x7FEB352D)).astype(np.uint32)
    x ^= x >> np.uint32(15)
    x = (x * np.uint32(0x846CA68B)).astype(np.uint32)
    x ^= x >> np.uint32(16)
    return x


def payload_bytes(seed: int, first_stream: int, n_streams: int, stride: int) -> np.ndarray:
    """uint8 [n_streams, stride]: byte j of stream s = hash(seed, s, j) & 0xFF."""
    with np.errstate(over="ignore"):
        s = (np.arange(first_stream, first_stream + n_streams, dtype=np.uint64)[:, None]
             * np.uint64(0x9E3779B1)).astype(np.uint32)
        j = np.arange(stride, dtype=np.uint32)[None, :]
        key = _hash32(np.uint32(seed & 0xFFFFFFFF) ^ _hash32(s))
        return (_hash32(key ^ (j * np.uint32(0x85EBCA6B))) & np.uint32(0xFF)).astype(np.uint8)


def frames_needed(bit_frames: int, ts_cycles: int, nbytes: int) -> int:
    """len(Transmitter.__getFrames(data)) for valid bauds (ref:452-469)."""
    return ts_cycles * 2 * bit_frames + 4 * bit_frames + 14 * nbytes * bit_frames + TAIL_SILENCE


def ts_cycles_for(baud: int, training_time: float = 0.5) -> int:
    return int(baud * training_time / 2)          # ref:438


def snr_to_scale_q24(snr_db: float) -> int:
    """Noise scale for afsk_add_noise_batch: sigma = 32767.5 / 10^(snr/20).

    The generator's raw sum has std sqrt(16 * (65536^2 - 1) / 12); the scale is
    rounded once on the host so CPU oracle and GPU use the identical integer.
    """
    sigma = 32767.5 / (10.0 ** (snr_db / 20.0))
    gen_std = (16.0 * (65536.0 ** 2 - 1.0) / 12.0) ** 0.5
    return int(round(sigma / gen_std * (1 << 24)))
