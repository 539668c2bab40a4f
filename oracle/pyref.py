"""Pure-Python scalar restatement of the receiver hot path -- TEST INFRASTRUCTURE ONLY.

Interpreter-speed twin of oracle/afsk_oracle.c: plain lists and integer arithmetic, no
numpy in the hot loops, so that `bench.py` can time a "reference-shaped" CPU figure on the
GPU box (the real reference, /root/reference/afskmodem.py, never travels there).  It is
written from the algorithm, function by function ("ref:" = afskmodem.py line numbers), and is
pinned by the same golden vectors as the C oracle (tests/test_oracle_golden.py).
"""
from __future__ import annotations

HI, LO = 32767, -32768
SYNC_WINDOW = 4096          # ref:323,327
DEAD_ZONE = 512             # ref:290-292


def tones(bit_frames: int):
    """(space, mark, training) templates for a valid bit_frames (ref:68-91)."""
    h, q = bit_frames // 2, bit_frames // 4
    space = [HI] * h + [LO] * h
    mark = ([HI] * q + [LO] * q) * 2
    return space, mark, mark + space


def mean_abs(frames) -> int:
    return int(sum(abs(v) for v in frames) / len(frames))            # ref:94-98


def mean_abs_diff(a, b) -> int:
    return int(sum(abs(x - y) for x, y in zip(a, b)) / len(a))        # ref:101-107


def limit(chunk):
    return [HI if v > DEAD_ZONE else (LO if v < -DEAD_ZONE else 0) for v in chunk]   # ref:287-296


def clock_index(frames, training) -> int:
    if len(frames) < SYNC_WINDOW:                                     # ref:323-325
        return -1
    n = len(training)
    best, best_i = None, 0
    for i in range(SYNC_WINDOW - n):                                  # ref:327-331
        d = mean_abs_diff(training, frames[i:i + n])
        if best is None or d < best:                                  # ref:332-337 first minimum
            best, best_i = d, i
    return best_i


def symbol_bit(chunk, mark, space) -> int:
    lim = limit(chunk)                                                # ref:344
    return 1 if mean_abs_diff(mark, lim) < mean_abs_diff(space, lim) else 0   # ref:346-351


def decode_bits(frames, bit_frames: int, amp_end: int = 14000):
    """-> (bits list, clock_idx, term_frame) as Receiver.__decodeBits (ref:354-381)."""
    space, mark, training = tones(bit_frames)
    i = clock_index(frames, training)
    if i < 0:
        return [], -1, -1
    ci = i
    window = [0, 0, 0, 0]                                             # ref:361
    last = len(frames) - bit_frames
    while i < last:                                                   # ref:362-366
        b = symbol_bit(frames[i:i + bit_frames], mark, space)
        i += bit_frames
        window = window[1:] + [b]
        if window == [1, 0, 0, 0]:                                    # ref:386-390
            break
    term = i
    bits = []
    while i < last:                                                   # ref:372-378
        chunk = frames[i:i + bit_frames]
        if mean_abs(chunk) < amp_end:
            break
        bits.append(symbol_bit(chunk, mark, space))
        i += bit_frames
    return bits, ci, term


_H = ((1, 0, 1, 0, 1, 0, 1), (0, 1, 1, 0, 0, 1, 1), (0, 0, 0, 1, 1, 1, 1))   # ref:125-129


def hamming_decode(bits):
    out = []
    for k in range(0, len(bits) - 6, 7):                              # ref:154-163
        r = list(bits[k:k + 7])
        s = [sum(h * x for h, x in zip(row, r)) % 2 for row in _H]    # ref:132-138, 146
        pos = s[2] * 4 + s[1] * 2 + s[0]                              # ref:147
        if pos:
            r[pos - 1] ^= 1                                           # ref:149-150
        out += [r[2], r[4], r[5], r[6]]                               # ref:151
    return out


def pack_bytes(bits) -> bytes:
    return bytes(int("".join(map(str, bits[k:k + 8])), 2) for k in range(0, len(bits) - 7, 8))   # ref:393-399


def demod(frames, bit_frames: int, amp_end: int = 14000):
    """frames (list[int]) -> (payload bytes, nbits, clock_idx, term_frame): ref:420-427."""
    bits, ci, term = decode_bits(frames, bit_frames, amp_end)
    return pack_bytes(hamming_decode(bits)), len(bits), ci, term
