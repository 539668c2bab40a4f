"""CPU checks (numpy, exhaustive where feasible) of the arithmetic identities the HIP kernels
rely on.  They restate the tricks, not the kernels: each test names the place in
afskmodem_amd/csrc that uses the identity, so a reader can see why the integer shortcuts are
exact.  The kernels themselves are checked against the oracle in tests/test_gpu_parity.py."""
import numpy as np
import pytest

BFS = (20, 40, 80, 160)


def training_cycle(bf):
    q, h = bf // 4, bf // 2
    mark = [32767 if ((j // q) & 1) == 0 else -32768 for j in range(bf)]       # ref:80-85
    space = [32767 if j < h else -32768 for j in range(bf)]                    # ref:68-77
    return np.array(mark + space, np.int64)                                    # ref:88-91


@pytest.mark.parametrize("bf", BFS)
def test_sliding_correlation_identity(bf):
    """afsk_demod_fast.h recover_clock_index_lanes: total(i) = 65535*bf + sum_j sigma_j x[i+j] and
    total(i+1) - total(i) = x[i] - 2x[i+q] + 2x[i+2q] - 2x[i+3q] + 2x[i+bf] - 2x[i+bf+h] + x[i+2bf]."""
    rng = np.random.default_rng(bf)
    x = rng.integers(-32768, 32768, 4096 + 8, dtype=np.int64)
    x[:64] = rng.choice([-32768, 32767, -513, -512, 512, 513, 0], 64)          # extremes
    tc = training_cycle(bf)
    n, q, h = 2 * bf, bf // 4, bf // 2
    noff = 4096 - n
    brute = np.array([np.abs(tc - x[i: i + n]).sum() for i in range(noff)])
    sigma = np.where(tc == 32767, -1, 1)
    direct = 65535 * bf + np.array([(sigma * x[i: i + n]).sum() for i in range(noff)])
    assert np.array_equal(brute, direct)
    i = np.arange(noff - 1)
    delta = (x[i] - 2 * x[i + q] + 2 * x[i + 2 * q] - 2 * x[i + 3 * q] + 2 * x[i + bf]
             - 2 * x[i + bf + h] + x[i + n])
    assert np.array_equal(np.diff(brute), delta)
    # the prefix-sum form of the first design (FLAGS & 8) is the same function
    p = np.concatenate([[0], np.cumsum(x)])
    i = np.arange(noff)
    pref = 65535 * bf + p[i] + p[i + n] + 2 * (p[i + 2 * q] + p[i + bf] - p[i + q] - p[i + 3 * q] - p[i + bf + h])
    assert np.array_equal(brute, pref)
    assert brute.max() < 2 ** 27                                               # magic-division range


@pytest.mark.parametrize("bf", BFS)
def test_first_minimum_of_truncated_mean_by_threshold(bf):
    """Pass 2 of the clock recovery: the reference keeps the FIRST index of the minimal
    int(total / n) (ref:332-337).  The kernel finds min(total), then the first index with
    total < (min // n + 1) * n -- the same index, with one division per stream."""
    n = 2 * bf
    rng = np.random.default_rng(100 + bf)
    for trial in range(200):
        lo = int(rng.integers(0, 65535 * n - 5 * n))
        totals = rng.integers(lo, lo + int(rng.integers(1, 6 * n)), 600)
        means = totals // n
        want = int(np.argmin(means))                                           # first minimum
        bound = (int(totals.min()) // n + 1) * n
        got = int(np.nonzero(totals < bound)[0][0])
        assert got == want


@pytest.mark.parametrize("n", [40, 80, 160, 320])
def test_mul_hi_magic_division_is_exact(n):
    """floor(m / n) = mul_hi(m, ceil(2^36 / n)) >> 4 for every m < 2^27 (afsk_demod_fast.h)."""
    M = ((1 << 36) + n - 1) // n
    assert M < (1 << 32)
    for lo in range(0, 1 << 27, 1 << 22):                                      # exhaustive, in 32 MB slabs
        mm = np.arange(lo, lo + (1 << 22), dtype=np.uint64)
        assert np.array_equal(((mm * np.uint64(M)) >> np.uint64(32)) >> np.uint64(4), mm // np.uint64(n))


def test_float_quarter_division_of_the_modulator_is_exact():
    """afsk_synth.hip tone_words: Q0 = (uint)(((float)x0 + 0.5f) * (1.0f / q)) == x0 // q for every
    x0 < 2^14 + 2048 and every quarter width q the kernel accepts for blocks of 16384 samples."""
    x0 = np.arange(0, 16384 + 2048, dtype=np.float32)
    for q in list(range(1, 512)):
        rcp = np.float32(1.0) / np.float32(q)
        got = ((x0 + np.float32(0.5)) * rcp).astype(np.uint32)
        assert np.array_equal(got, (x0.astype(np.uint32) // q)), q


def test_small_quarter_reciprocal_multiply_is_exact():
    """afsk_synth.hip tone_words (q < 8): ((r0 + j) * ceil(65536 / q)) >> 16 == (r0 + j) // q
    for r0 + j <= 13."""
    for q in range(1, 8):
        mq = (65536 + q - 1) // q
        v = np.arange(0, 14)
        assert np.array_equal((v * mq) >> 16, v // q)


def test_limiter_sad_identity():
    """Phase B: SAD of the limited samples against a lo template = 65535*n - SAD against the hi
    template, so quarter sums h0..h3 against 'hi' give both correlators (afsk_demod_fast.h)."""
    rng = np.random.default_rng(7)
    for bf in BFS:
        q = bf // 4
        x = rng.integers(-2000, 2000, bf)
        x[:6] = (513, 512, -512, -513, 32767, -32768)
        lim = np.where(x > 512, 32767, np.where(x < -512, -32768, 0))         # ref:287-296
        tc = training_cycle(bf)
        mark_t, space_t = tc[:bf], tc[bf:]
        mark = np.abs(mark_t - lim).sum()
        space = np.abs(space_t - lim).sum()
        hq = [np.abs(32767 - lim[k * q: (k + 1) * q]).sum() for k in range(4)]
        full = 65535 * q
        assert mark == 2 * full + hq[0] - hq[1] + hq[2] - hq[3]
        assert space == 2 * full + hq[0] + hq[1] - hq[2] - hq[3]


def _compress(x, lps):
    if lps == 2:
        x &= 0x5555555555555555
        x = (x | (x >> 1)) & 0x3333333333333333
        x = (x | (x >> 2)) & 0x0F0F0F0F0F0F0F0F
        x = (x | (x >> 4)) & 0x00FF00FF00FF00FF
        x = (x | (x >> 8)) & 0x0000FFFF0000FFFF
        x = (x | (x >> 16)) & 0x00000000FFFFFFFF
    else:
        x &= 0x1111111111111111
        x = (x | (x >> 3)) & 0x0303030303030303
        x = (x | (x >> 6)) & 0x000F000F000F000F
        x = (x | (x >> 12)) & 0x000000FF000000FF
        x = (x | (x >> 24)) & 0x000000000000FFFF
    return x


def test_scalar_bit_compaction():
    """compress_bits<LPS> (afsk_demod_fast.h): bit j of the result = bit j*LPS of the ballot."""
    rng = np.random.default_rng(3)
    for lps in (2, 4):
        for _ in range(500):
            x = int(rng.integers(0, 1 << 63)) | (int(rng.integers(0, 2)) << 63)
            want = sum(((x >> (j * lps)) & 1) << j for j in range(64 // lps))
            assert _compress(x, lps) == want


def test_quarter_bitmap_expansion_of_the_modulator():
    """afsk_synth.hip: 8 symbol kinds (1 = mark) -> 32 quarter bits, nibble 0b0101 for a mark
    (hi,lo,hi,lo) and 0b0011 for a space (hi,hi,lo,lo); bit 0 = first quarter."""
    for kinds in range(256):
        x = kinds
        x = (x | (x << 12)) & 0x000F000F
        x = (x | (x << 6)) & 0x03030303
        x = (x | (x << 3)) & 0x11111111
        word = 0x33333333 ^ (x * 6)
        for k in range(8):
            nib = (word >> (4 * k)) & 15
            assert nib == (0b0101 if (kinds >> k) & 1 else 0b0011)


def test_hamming_popcount_syndrome():
    """hamming_syndrome / hamming_nibble (afsk_demod_impl.h) against the matrix form ref:125-151."""
    H = [[1, 0, 1, 0, 1, 0, 1], [0, 1, 1, 0, 0, 1, 1], [0, 0, 0, 1, 1, 1, 1]]
    for cw in range(128):
        r = [(cw >> t) & 1 for t in range(7)]
        syn = [sum(H[a][j] * r[j] for j in range(7)) % 2 for a in range(3)]
        pos = syn[2] * 4 + syn[1] * 2 + syn[0]
        s0 = bin(cw & 0x55).count("1") & 1
        s1 = bin(cw & 0x66).count("1") & 1
        s2 = bin(cw & 0x78).count("1") & 1
        assert pos == s2 * 4 + s1 * 2 + s0
