"""CPU checks (numpy, exhaustive where feasible) of the arithmetic identities the HIP kernels
rely on.  They restate the tricks, not the kernels: each test names the place in
afskmodem_amd/csrc that uses the identity, so a reader can see why the integer shortcuts are
exact.  The kernels themselves are checked against the oracle in tests/test_gpu_*.py."""
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
    """afsk_demod_sync.h recover_clock_index_lanes: total(i) = 65535*bf + sum_j sigma_j x[i+j] and
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
    """floor(m / n) = mul_hi(m, ceil(2^36 / n)) >> 4 for every m < 2^27 (afsk_demod_sync.h)."""
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
    template, so quarter sums h0..h3 against 'hi' give both correlators (afsk_demod_phasec.h / afsk_demod_rounds_*.h)."""
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
    """compress_bits<LPS> (afsk_demod_phasec.h): bit j of the result = bit j*LPS of the ballot."""
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


# ---- general-piece geometry (afsk_demod_rounds_gp.h GpGeom / gp_rounds) -----------------------------------
GP_BFS = (128, 192, 200, 300, 384, 400, 500, 600, 640, 800, 960, 1000, 1200, 1500, 1600, 1920, 2000)


def gp_geom(bf):
    """GpGeom<BF> restated: lanes per symbol, and per lane (quarter k, first dword, dword count)."""
    q = bf // 4
    lps = 4
    while lps < 64 and (64 // lps) * 2 * bf > 8192:
        lps *= 2
    lpq = lps // 4
    qs = lambda k: (k * q + 1) // 2  # noqa: E731
    lanes = []
    for part in range(lps):
        k, j = part // lpq, part % lpq
        nk = qs(k + 1) - qs(k)
        d0, d1 = qs(k) + (j * nk) // lpq, qs(k) + ((j + 1) * nk) // lpq
        lanes.append((k, d0, d1 - d0))
    return lps, lanes


@pytest.mark.parametrize("bf", GP_BFS)
def test_general_piece_map_covers_every_symbol_once(bf):
    """gp_rounds: the LPS lanes of a symbol own disjoint runs of its bf/2 dwords, every lane NB + 1 or
    NB + 2 of them, a run lies inside ONE quarter except that its last dword may straddle into the next
    (odd quarter length), rounds are whole symbols of at most 8 KiB, a piece fits the 256-byte mirror."""
    lps, lanes = gp_geom(bf)
    q, d = bf // 4, bf // 2
    assert 64 % lps == 0 and (64 // lps) * 2 * bf <= 8192
    cover = np.zeros(d, int)
    sizes = {n for _, _, n in lanes}
    nb = min(sizes) - 1
    assert nb >= 1 and sizes <= {nb + 1, nb + 2} and 4 * (nb + 2) + 4 <= 256
    for k, d0, n in lanes:
        cover[d0: d0 + n] += 1
        first, last = 2 * d0, 2 * (d0 + n) - 1                  # first / last sample of the piece
        assert first // q == k                                  # the piece starts in its quarter
        assert (last - 1) // q == k                             # every sample but the very last one is in it
        assert last // q in (k, k + 1) and last < bf
    assert (cover == 1).all()


@pytest.mark.parametrize("bf", GP_BFS)
def test_general_piece_sums_equal_the_symbol_correlators(bf):
    """gp_rounds arithmetic: NB dwords against the lane's constant template via ONE SAD against "hi"
    (SAD against lo = 65535 * n - SAD against hi) + two tail slots with per-lane template dwords, summed
    over the lanes, equals the mark and space SADs of ref:346-347 over the limited symbol."""
    rng = np.random.default_rng(bf)
    q, h = bf // 4, bf // 2
    mark_t = np.array([65535 if ((j // q) & 1) == 0 else 0 for j in range(bf)])     # biased templates
    space_t = np.array([65535 if j < h else 0 for j in range(bf)])
    lps, lanes = gp_geom(bf)
    nb = min(n for _, _, n in lanes) - 1
    for _ in range(5):
        lim = rng.choice([0, 0x8000, 0xFFFF], bf)                                  # limited samples, biased
        want_m, want_s = np.abs(mark_t - lim).sum(), np.abs(space_t - lim).sum()
        got_m = got_s = 0
        for k, d0, n in lanes:
            x = lim[2 * d0: 2 * (d0 + n)]
            hsum = np.abs(65535 - x[: 2 * nb]).sum()
            got_m += hsum if k % 2 == 0 else 65535 * 2 * nb - hsum
            got_s += hsum if k < 2 else 65535 * 2 * nb - hsum
            for dd in range(nb, n):                                                # tail slots A (and B)
                lo, hi = 2 * (d0 + dd), 2 * (d0 + dd) + 1
                is_last = dd == n - 1
                kl = hi // q if is_last else k                                      # only the last dword can straddle
                tm = (65535 if k % 2 == 0 else 0, 65535 if kl % 2 == 0 else 0)
                ts = (65535 if k < 2 else 0, 65535 if kl < 2 else 0)
                got_m += abs(tm[0] - lim[lo]) + abs(tm[1] - lim[hi])
                got_s += abs(ts[0] - lim[lo]) + abs(ts[1] - lim[hi])
        assert (got_m, got_s) == (want_m, want_s)


@pytest.mark.parametrize("bf", (240, 300, 500, 1500, 2000))
def test_sliding_correlation_identity_long_and_odd_quarters(bf):
    """recover_clock_index_lane_steps for long templates: the 7-tap delta holds for any quarter length
    (also odd ones: 75, 125, 375 samples), and floor(m / n) by the 2^36 magic multiplier is exact while
    65535 * n^2 < 2^36 (n = 2 * bf <= 960); longer templates use the float estimate + fix-up."""
    rng = np.random.default_rng(bf)
    x = rng.integers(-32768, 32768, 4096 + 8, dtype=np.int64)
    tc = training_cycle(bf)
    n, q, h = 2 * bf, bf // 4, bf // 2
    noff = 4096 - n
    brute = np.array([np.abs(tc - x[i: i + n]).sum() for i in range(noff)])
    i = np.arange(noff - 1)
    delta = (x[i] - 2 * x[i + q] + 2 * x[i + 2 * q] - 2 * x[i + 3 * q] + 2 * x[i + bf]
             - 2 * x[i + bf + h] + x[i + n])
    assert np.array_equal(np.diff(brute), delta)
    if 65535 * n * n < 2 ** 36:
        mm = -(-2 ** 36 // n)
        m = np.concatenate([rng.integers(0, 65535 * n + 1, 20000), [0, 65535 * n, n - 1, n, 65535 * n - 1]]).astype(object)
        assert all(((int(v) * mm) >> 36) == int(v) // n for v in m)


def test_bit_frames_4_decision_is_a_level_comparison():
    """multi_rounds<4> (12000 baud): with one sample per quarter the decision int(mark / 4) < int(space / 4)
    of ref:346-351 equals L1 < L2 on the limited samples -- exhaustive over the 3^4 level combinations."""
    import itertools
    levels = (-32768, 0, 32767)                                  # limiter outputs (ref:287-296)
    mark_t = (32767, -32768, 32767, -32768)                      # ref:80-85 at bit_frames 4
    space_t = (32767, 32767, -32768, -32768)                     # ref:68-77
    for s in itertools.product(levels, repeat=4):
        md = int(sum(abs(t - v) for t, v in zip(mark_t, s)) / 4)
        sd = int(sum(abs(t - v) for t, v in zip(space_t, s)) / 4)
        assert (md < sd) == (s[1] < s[2]), s


def test_two_codeword_bit_sliced_hamming_decode():
    """afsk_demod_phasec.h hamming_byte (r5): both Hamming(7,4) codewords of an output byte decoded in the same registers --
    parity checks bit-sliced through c ^ (c >> 4) and c ^ (c >> 1), error positions at bits 0-2 / 7-9, v_bfrev to line
    the data bits up.  All 2^14 received words against the per-codeword syndrome decode of ref:145-151, plus the funnel
    shift that takes the 14 bits out of two neighbouring dwords of the bit buffer."""
    def ref_nibble(cw):                      # bit t of cw = received bit t (ref:146-151)
        s0 = bin(cw & 0x55).count("1") & 1
        s1 = bin(cw & 0x66).count("1") & 1
        s2 = bin(cw & 0x78).count("1") & 1
        pos = s2 * 4 + s1 * 2 + s0
        if pos:
            cw ^= 1 << (pos - 1)
        return ((cw >> 2) & 1) << 3 | ((cw >> 4) & 1) << 2 | ((cw >> 5) & 1) << 1 | ((cw >> 6) & 1), pos

    def bitrev32(x):
        return int(format(x & 0xFFFFFFFF, "032b")[::-1], 2)

    for c in range(1 << 14):
        t1, t2 = c ^ (c >> 4), c ^ (c >> 1)
        a = t1 >> 2
        s0, s1, s2 = t1 ^ a, (t1 >> 1) ^ a, (t2 >> 3) ^ (t2 >> 5)
        pos2 = (s0 & 0x81) | ((s1 & 0x81) << 1) | ((s2 & 0x81) << 2)
        p0, p1 = pos2 & 7, (pos2 >> 7) & 7
        fixed = c ^ ((1 << p0) >> 1) ^ (((1 << p1) >> 1) << 7)
        rev = bitrev32(fixed)
        hi = ((rev >> 25) & 7) | ((rev >> 26) & 8)
        lo = ((rev >> 18) & 7) | ((rev >> 19) & 8)
        n0, q0 = ref_nibble(c & 127)
        n1, q1 = ref_nibble(c >> 7)
        assert (hi, lo, p0, p1) == (n0, n1, q0, q1), c
        assert (pos2 >> 7) == p1                                      # what the kernel tests for "second codeword corrected"
    # the funnel shift: 14 bits from bit g of a 4096-bit circular buffer of dwords
    rng = np.random.default_rng(14)
    bits = rng.integers(0, 2, 4096)
    dw = [int(sum(int(bits[32 * i + k]) << k for k in range(32))) for i in range(128)]
    for g in list(range(0, 4096, 7)) + [4095, 4090, 4083]:
        w0, w1 = dw[(g >> 5) & 127], dw[((g >> 5) + 1) & 127]
        c = (((w1 << 32) | w0) >> (g & 31)) & 0x3FFF                 # v_alignbit_b32(w1, w0, g & 31) & 0x3FFF
        want = sum(int(bits[(g + k) % 4096]) << k for k in range(14))
        assert c == want, g
