// afsk_demod_phasec.h -- part of the single-pass demodulator (afsk_demod_fast.h includes the parts in order; see its header
// comment for the overall design): phase B helpers (SAD against the "hi" template) and phase C: terminator scan, squelch stop, the deferred
// vectorised Hamming decode + byte pack (rxd_pass / rxd_round / rxd_flush), DPP group sums, ballot compaction.
#pragma once

namespace afsk {

// ------------------------------------------------------------------ phase B (fast)
// Sum over the dwords [D0, D1) of |0xFFFF - limited(x)| per 16-bit half: the SAD of the
// limited samples against a "hi" (32767) template.  Against a "lo" (-32768) template the
// SAD of the same samples is 65535 * n_samples minus this, so one v_sad_u16 per dword
// serves both the mark and the space correlator (ref:346-347).
template <int D0, int D1>
__device__ __forceinline__ uint32_t hi_sad(const uint32_t (&x)[20]) {
    uint32_t h = 0;
#pragma unroll
    for (int d = D0; d < D1; d++) h = __builtin_amdgcn_sad_u16(limit_pair_biased(x[d]), 0xFFFFFFFFu, h);
    return h;
}

template <int D0, int D1>
__device__ __forceinline__ uint32_t quiet_sum(const uint32_t (&x)[20]) {   // 32768 n - sum|x| (ref:94-98; see quiet_sad)
    uint32_t a = 0;
#pragma unroll
    for (int d = D0; d < D1; d++) a = quiet_sad(x[d], a);
    return a;
}

// ---- phase C for the single-pass kernel: deferred Hamming decode -------------------------
// Every pass only (a) scans for the training terminator, (b) once in the data phase checks the
// squelch stop, and (c) parks its symbol decisions in a small circular bit buffer in LDS (bit
// g % 64 of word (g / 64) % kBitWords = decision of symbol g).  The ECC decode + byte pack
// (ref:145-163, 393-399) runs vectorised, one lane per output byte, whenever 64 bytes are
// ready and once at the end -- instead of ~100 dependent scalar instructions per pass.
constexpr int kBitWords = 64;                                   // 4096 symbols of history

struct RxDeferred {
    RxState st;               // phase / hist / term_sym as in the per-pass state machine
    int32_t end_sym;          // first symbol index past the data (valid once st.phase == 2)
    int32_t bytes_done;       // decoded bytes already stored
    int32_t filled;           // symbols parked so far (a multiple of the pass size)
    uint64_t cur;             // bits of the 64-symbol word being filled
};

__device__ __forceinline__ void rxd_init(RxDeferred& d) {
    d.st.phase = 0; d.st.hist = 0; d.st.nbits = 0; d.st.nbytes = 0; d.st.term_sym = -1;
    d.st.pend = 0; d.st.npend = 0; d.st.corrected = 0;
    d.end_sym = 0; d.bytes_done = 0; d.filled = 0; d.cur = 0;
}

// squelch stop inside a pass whose data symbols start at `start` (ref:372-376)
__device__ __forceinline__ void rxd_stop(RxDeferred& d, uint64_t amp_ok, int start, int nv, int k0) {
    const uint64_t quiet = ~amp_ok;
    if (quiet == 0) return;           // every symbol loud: all data passes of a stream but its last (r5: two scalar instructions instead of ten)
    const uint64_t valid = nv >= 64 ? ~0ull : ((1ull << nv) - 1ull);
    const uint64_t stop = valid & ~((1ull << start) - 1ull) & quiet;       // start < 64
    if (stop) {
        d.end_sym = k0 + __builtin_ctzll(stop);
        d.st.phase = 2;
    }
}

// park the PS decisions of the pass that starts at symbol k0 (k0 % PS == 0, 64 % PS == 0)
template <int PS>
__device__ __forceinline__ void rxd_store(RxDeferred& d, uint64_t bits, int nv, int k0, int lane,
                                          unsigned long long* words) {
    const uint64_t valid = nv >= 64 ? ~0ull : ((1ull << nv) - 1ull);
    d.filled = k0 + PS;
    if constexpr (PS == 64) {
        if (lane == 0) words[(k0 >> 6) & (kBitWords - 1)] = bits & valid;
    } else {
        d.cur |= (bits & valid) << (k0 & 63);
        if (((k0 & 63) + PS) == 64) {
            if (lane == 0) words[(k0 >> 6) & (kBitWords - 1)] = d.cur;
            d.cur = 0;
        }
    }
}

// Both Hamming(7,4) codewords of one output byte at once (r5; ref:145-163, 393-399).  c = 14 received bits, bit t of
// the first codeword at bit t, of the second at bit 7 + t.  The three parity checks of ref:126-128 (rows 1010101 /
// 0110011 / 0001111) are formed bit-sliced for both codewords in the same registers: with t1 = c ^ (c >> 4) and
// t2 = c ^ (c >> 1)
//     s0 = c0^c2^c4^c6 = bit 0 of t1 ^ (t1 >> 2),   s1 = c1^c2^c5^c6 = bit 0 of (t1 >> 1) ^ (t1 >> 2),
//     s2 = c3^c4^c5^c6 = bit 0 of (t2 >> 3) ^ (t2 >> 5)                 (second codeword: the same at bit 7)
// (the bits of c that the shifts drag across the codeword boundary only reach positions that are not read).  The
// error position s2 s1 s0 flips its bit (ref:149-150), and v_bfrev lines the data bits d1..d4 = bits 2, 4, 5, 6 up
// in the order the nibble wants them (ref:151).  ~35 instructions per byte instead of ~60 for two separate
// popcount decodes -- at 12000 baud a stream has six flushes of 64 bytes, the largest item of what it executes
// beyond a 1200-baud stream (exhaustive check of the identity: tests/test_kernel_math.py).
// Returns the decoded byte; pos2 = the two error positions (first codeword bits 0-2, second bits 7-9; 0 = clean).
__device__ __forceinline__ uint32_t hamming_byte(uint32_t c, uint32_t& pos2) {
    const uint32_t t1 = c ^ (c >> 4), t2 = c ^ (c >> 1);
    const uint32_t a = t1 >> 2;
    const uint32_t s0 = t1 ^ a, s1 = (t1 >> 1) ^ a, s2 = (t2 >> 3) ^ (t2 >> 5);
    pos2 = (s0 & 0x81u) | ((s1 & 0x81u) << 1) | ((s2 & 0x81u) << 2);            // ref:147, both codewords
    const uint32_t p0 = pos2 & 7u, p1 = (pos2 >> 7) & 7u;
    const uint32_t fixed = c ^ ((1u << p0) >> 1) ^ (((1u << p1) >> 1) << 7);    // ref:149-150 (position 0: nothing to flip)
    const uint32_t rev = __builtin_bitreverse32(fixed);                         // bit k -> bit 31 - k
    const uint32_t hi = ((rev >> 25) & 7u) | ((rev >> 26) & 8u);                // d1 d2 d3 d4 of the first codeword
    const uint32_t lo = ((rev >> 18) & 7u) | ((rev >> 19) & 8u);                // ... of the second
    return (hi << 4) | lo;                                                      // ref:393-399: first nibble high
}

// decode and store every byte whose 14 coded bits lie below symbol index `avail`
template <int PS>
__device__ __forceinline__ void rxd_flush(RxDeferred& d, int avail, int lane,
                                          unsigned long long* words, uint8_t* out_row, int out_stride) {
    if (d.st.term_sym < 0 || avail <= d.st.term_sym) return;
    const int jnew = (avail - d.st.term_sym) / 14;
    if (jnew <= d.bytes_done) return;
    if constexpr (PS != 64) {         // a partly filled word is not in LDS yet
        if ((d.filled & 63) != 0 && lane == 0) words[(d.filled >> 6) & (kBitWords - 1)] = d.cur;
    }
    wave_lds_sync();                  // lane 0 stored the words, every lane reads them
    const uint32_t* dwords = reinterpret_cast<const uint32_t*>(words);          // the bit buffer as 128 dwords
    for (int j0 = d.bytes_done; j0 < jnew; j0 += 64) {
        const int j = j0 + lane;
        bool fix0 = false, fix1 = false;       // soft output: non-zero syndromes (ref:147)
        if (j < jnew) {
            // the 14 coded bits start at bit g of the circular buffer: two neighbouring dwords and one funnel shift
            const int g = d.st.term_sym + 14 * j;
            const uint32_t w0 = dwords[(g >> 5) & (2 * kBitWords - 1)];
            const uint32_t w1 = dwords[((g >> 5) + 1) & (2 * kBitWords - 1)];
            const uint32_t c = __builtin_amdgcn_alignbit(w1, w0, (uint32_t)g & 31u) & 0x3FFFu;
            uint32_t pos2;
            const uint32_t byte = hamming_byte(c, pos2);
            if (j < out_stride) out_row[j] = (uint8_t)byte;
            fix0 = (pos2 & 7u) != 0;
            fix1 = (pos2 >> 7) != 0;
        }
        d.st.corrected += (int32_t)__popcll(__ballot(fix0)) + (int32_t)__popcll(__ballot(fix1));
    }
    wave_lds_sync();                  // later passes overwrite old words
    d.bytes_done = jnew;
}

// 64 coded symbols per flush: (avail - term_sym) / 14 - bytes_done >= 64 without the division (r5: every scalar
// instruction of a round costs the wave four cycles of its issue slot, like a vector one)
__device__ __forceinline__ bool rxd_flush_due(const RxDeferred& d, int avail) {
    return d.st.phase == 1 && avail - d.st.term_sym >= 14 * (d.bytes_done + 64);
}

// one pass of PS symbols: terminator scan, lazy squelch amplitude, park the bits, maybe flush
template <int PS, class AmpFn>
__device__ __forceinline__ void rxd_pass(RxDeferred& d, uint64_t bmask, int nv, int k0, int lane,
                                         unsigned long long* words, uint8_t* out_row, int out_stride,
                                         AmpFn&& amp_ok_mask) {
    int start = -1;
    if (d.st.phase == 0) {
        // A terminator (1,0,0,0: ref:386-390) needs zero decisions in a row, which the training tone -- alternating
        // decisions -- does not have: when no position of a FULL pass holds a zero right behind a zero (the
        // decision before the pass included) only the three-decision history moves on; everything else takes
        // the complete scan of rx_training.
        const uint64_t prev = (bmask << 1) | ((d.st.hist >> 2) & 1u);            // the decision before each one
        constexpr uint64_t kAll = PS >= 64 ? ~0ull : ((1ull << (PS & 63)) - 1ull);
        if (nv == PS && PS >= 3 && ((~(bmask | prev)) & kAll) == 0)
            d.st.hist = (uint32_t)(bmask >> (PS - 3)) & 7u;
        else
            start = rx_training(d.st, bmask, nv, k0);
    } else if (d.st.phase == 1) {
        start = 0;
    }
    if (start >= 0 && start < nv) rxd_stop(d, amp_ok_mask(), start, nv, k0);
    rxd_store<PS>(d, bmask, nv, k0, lane, words);
    if (rxd_flush_due(d, k0 + nv)) rxd_flush<PS>(d, k0 + nv, lane, words, out_row, out_stride);
}

// Lane p <- the wave-uniform 64-bit word B[p] (p < SPL), zero in every other lane: one v_writelane_b32 per half
// instead of a v_mov + v_cndmask pair.  ONE asm statement for all of them, opened by `s_nop 1`: the words are
// ballots, i.e. SGPRs (or VCC) written by VALU compares, and on gfx940 / gfx950 a VALU instruction that reads an
// SGPR needs two wait states behind the VALU instruction that wrote it.  The compiler pads its own code for that
// (its hazard recogniser) but cannot see into inline asm -- separate statements, scheduled right behind their
// compares, read stale words (r5: every 2400-baud stream found a terminator that was not there).  The lane select
// is an immediate, so the ISA's other hazard of this instruction (SGPR lane select written by VALU) cannot arise.
#define AFSK_WL(p, lo, hi) "\n\tv_writelane_b32 %0, %" #lo ", " #p "\n\tv_writelane_b32 %1, %" #hi ", " #p
template <int SPL>
__device__ __forceinline__ void spread_words(const uint64_t (&B)[SPL], uint32_t& wlo, uint32_t& whi) {
    uint32_t lo = 0, hi = 0;
#define AFSK_S(p) "s"((uint32_t)B[p]), "s"((uint32_t)(B[p] >> 32))
    if constexpr (SPL == 2) {
        asm volatile("s_nop 1" AFSK_WL(0, 2, 3) AFSK_WL(1, 4, 5) : "+v"(lo), "+v"(hi) : AFSK_S(0), AFSK_S(1));
    } else if constexpr (SPL == 3) {
        asm volatile("s_nop 1" AFSK_WL(0, 2, 3) AFSK_WL(1, 4, 5) AFSK_WL(2, 6, 7)
                     : "+v"(lo), "+v"(hi) : AFSK_S(0), AFSK_S(1), AFSK_S(2));
    } else if constexpr (SPL == 4) {
        asm volatile("s_nop 1" AFSK_WL(0, 2, 3) AFSK_WL(1, 4, 5) AFSK_WL(2, 6, 7) AFSK_WL(3, 8, 9)
                     : "+v"(lo), "+v"(hi) : AFSK_S(0), AFSK_S(1), AFSK_S(2), AFSK_S(3));
    } else if constexpr (SPL == 5) {
        asm volatile("s_nop 1" AFSK_WL(0, 2, 3) AFSK_WL(1, 4, 5) AFSK_WL(2, 6, 7) AFSK_WL(3, 8, 9) AFSK_WL(4, 10, 11)
                     : "+v"(lo), "+v"(hi) : AFSK_S(0), AFSK_S(1), AFSK_S(2), AFSK_S(3), AFSK_S(4));
    } else if constexpr (SPL == 6) {
        asm volatile("s_nop 1" AFSK_WL(0, 2, 3) AFSK_WL(1, 4, 5) AFSK_WL(2, 6, 7) AFSK_WL(3, 8, 9) AFSK_WL(4, 10, 11) AFSK_WL(5, 12, 13)
                     : "+v"(lo), "+v"(hi) : AFSK_S(0), AFSK_S(1), AFSK_S(2), AFSK_S(3), AFSK_S(4), AFSK_S(5));
    } else if constexpr (SPL == 8) {
        asm volatile("s_nop 1" AFSK_WL(0, 2, 3) AFSK_WL(1, 4, 5) AFSK_WL(2, 6, 7) AFSK_WL(3, 8, 9) AFSK_WL(4, 10, 11) AFSK_WL(5, 12, 13)
                     AFSK_WL(6, 14, 15) AFSK_WL(7, 16, 17)
                     : "+v"(lo), "+v"(hi) : AFSK_S(0), AFSK_S(1), AFSK_S(2), AFSK_S(3), AFSK_S(4), AFSK_S(5), AFSK_S(6), AFSK_S(7));
    } else if constexpr (SPL == 10) {
        asm volatile("s_nop 1" AFSK_WL(0, 2, 3) AFSK_WL(1, 4, 5) AFSK_WL(2, 6, 7) AFSK_WL(3, 8, 9) AFSK_WL(4, 10, 11)
                     AFSK_WL(5, 12, 13) AFSK_WL(6, 14, 15) AFSK_WL(7, 16, 17) AFSK_WL(8, 18, 19) AFSK_WL(9, 20, 21)
                     : "+v"(lo), "+v"(hi)
                     : AFSK_S(0), AFSK_S(1), AFSK_S(2), AFSK_S(3), AFSK_S(4), AFSK_S(5), AFSK_S(6), AFSK_S(7), AFSK_S(8), AFSK_S(9));
    } else {
        const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
#pragma unroll
        for (int p = 0; p < SPL; p++) {
            lo = lane == p ? (uint32_t)B[p] : lo;              // v_cndmask with the scalar word as a source
            hi = lane == p ? (uint32_t)(B[p] >> 32) : hi;
        }
    }
#undef AFSK_S
    wlo = lo;
    whi = hi;
}
#undef AFSK_WL

// One ROUND of SPL x 64 symbols in a single phase-C step (instead of SPL dependent scalar passes):
// lane p < SPL takes the ballot word of symbols k0 + 64p .. k0 + 64p + 63 (v_cndmask), so the
// terminator scan (ref:386-390) and the squelch stop (ref:372-376) run on all SPL words at once in
// 64-bit VALU arithmetic -- the three decisions before a word come from the neighbouring lane by
// DPP row_shr:1 (lane 0: the carried history) -- and one ds_write_b64 parks all words.  What is
// left on the scalar unit is "any hit?" (one ballot) and, once per stream each, locating the first
// terminator / first quiet symbol.  amp_word(p) returns the "loud enough" ballot of slice p and is
// only evaluated from the round with the terminator on, like the reference (ref:361-366).
// all_loud() is this lane's "every one of my SPL symbols is loud enough" (the largest of their quiet sums against the
// threshold): only a round in which some lane says no -- the stream's last, normally -- forms the SPL amplitude words,
// spreads them over the lanes and locates the first quiet symbol (r5: 27 instead of 70 instructions per data round at
// 12000 baud; symbols before the data or past the end can only send a round down the exact path, never past it).
template <int SPL, class AmpFn, class AllLoudFn>
__device__ __forceinline__ void rxd_round(RxDeferred& d, const uint64_t (&B)[SPL], int32_t K, int k0,
                                          int lane, unsigned long long* words, uint8_t* out_row,
                                          int out_stride, AmpFn&& amp_word, AllLoudFn&& all_loud) {
    static_assert(SPL >= 2 && SPL <= 16, "one DPP row");
    // word p of the round goes to lane p: one v_writelane_b32 per half (r5; r4 moved every half through a
    // v_mov + v_cndmask pair -- 4 * SPL instructions per round, as many as the decisions themselves at 12000 baud)
    uint32_t wlo, whi;
    spread_words<SPL>(B, wlo, whi);
    // symbols of this lane's word that exist: all 64 in every round but the stream's last (wave-uniform test)
    uint64_t valid;
    if (K - k0 >= 64 * SPL) {
        valid = lane < SPL ? ~0ull : 0ull;
    } else {
        const int rem = K - k0 - 64 * lane;
        valid = (lane >= SPL || rem <= 0) ? 0ull : (rem >= 64 ? ~0ull : ((1ull << rem) - 1ull));
    }
    const uint64_t w = (((uint64_t)whi << 32) | wlo) & valid;
    if (lane < SPL) words[((k0 >> 6) + lane) & (kBitWords - 1)] = w;
    d.filled = k0 + 64 * SPL;
    const int nv = (K - k0) < 64 * SPL ? (K - k0) : 64 * SPL;  // symbols of this round
    int start = -1;                                            // first data symbol of the round, -1 = none
    if (d.st.phase == 0) {
        uint32_t phi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(w >> 32), 0x111, 0xf, 0xf, true);   // row_shr:1
        if (lane == 0) phi = d.st.hist << 29;                  // decisions k0-3 .. k0-1
        const uint64_t b1 = (w << 1) | (uint64_t)(phi >> 31);
        // no zero decision right behind a zero one anywhere in the round (the training tone alternates): no
        // terminator -- one 64-bit shift and one ballot instead of three shifts and the scan
        uint64_t any = __ballot((~(w | b1) & valid) != 0);
        uint64_t hit = 0;
        if (any) {
            const uint64_t b3 = (w << 3) | (uint64_t)(phi >> 29);
            const uint64_t b2 = (w << 2) | (uint64_t)(phi >> 30);
            hit = b3 & ~b2 & ~b1 & ~w & valid;                 // window == 1,0,0,0 (ref:386-390)
            any = __ballot(hit != 0);
        }
        if (any) {
            const int p = __builtin_ctzll(any);
            const uint64_t hw = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(hit >> 32), p) << 32) |
                                (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)hit, p);
            start = 64 * p + __builtin_ctzll(hw) + 1;
            d.st.term_sym = k0 + start;
            d.st.phase = 1;
        } else {
            d.st.hist = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(w >> 32), SPL - 1) >> 29;
        }
    } else if (d.st.phase == 1) {
        start = 0;
    }
    if (start >= 0 && start < nv && __ballot(!all_loud()) != 0) {          // squelch stop (ref:372-376)
        uint64_t A[SPL];
#pragma unroll
        for (int p = 0; p < SPL; p++) A[p] = amp_word(p);
        uint32_t alo, ahi;
        spread_words<SPL>(A, alo, ahi);
        const int rel = start - 64 * lane;                     // data starts at bit rel of this lane's word
        const uint64_t from = rel <= 0 ? ~0ull : (rel >= 64 ? 0ull : ~((1ull << rel) - 1ull));
        const uint64_t stop = ~(((uint64_t)ahi << 32) | alo) & valid & from;
        const uint64_t any = __ballot(stop != 0);
        if (any) {
            const int p = __builtin_ctzll(any);
            const uint64_t sw = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(stop >> 32), p) << 32) |
                                (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)stop, p);
            d.end_sym = k0 + 64 * p + __builtin_ctzll(sw);
            d.st.phase = 2;
        }
    }
    if (rxd_flush_due(d, k0 + nv)) rxd_flush<64>(d, k0 + nv, lane, words, out_row, out_stride);
}

// end of stream: K symbols were examined unless the squelch stopped earlier
template <int PS>
__device__ __forceinline__ void rxd_finish(RxDeferred& d, int32_t K, int lane, unsigned long long* words,
                                           uint8_t* out_row, int out_stride) {
    if (d.st.term_sym < 0) { d.st.nbits = 0; d.st.nbytes = 0; return; }
    const int end = d.st.phase == 2 ? d.end_sym : K;
    d.st.nbits = end > d.st.term_sym ? end - d.st.term_sym : 0;
    d.st.nbytes = d.st.nbits / 14;
    rxd_flush<PS>(d, end, lane, words, out_row, out_stride);
    if ((d.st.nbits / 7) & 1) {       // ECC.decode also corrects an odd last codeword (ref:157-162)
        if constexpr (PS != 64) {
            if ((d.filled & 63) != 0 && lane == 0) words[(d.filled >> 6) & (kBitWords - 1)] = d.cur;
        }
        wave_lds_sync();
        const int g = d.st.term_sym + 14 * d.st.nbytes;
        const uint64_t lo = words[(g >> 6) & (kBitWords - 1)];
        const uint64_t hi = words[((g >> 6) + 1) & (kBitWords - 1)];
        const int sh = g & 63;
        uint32_t c = (uint32_t)(lo >> sh);
        if (sh > 57) c |= (uint32_t)(hi << (64 - sh));
        d.st.corrected += hamming_syndrome(c & 127u) != 0;
    }
}

// Sum of a value over the 2, 4, 8 or 16 lanes of an aligned group, result in every lane of it (DPP only).
template <int LPS>
__device__ __forceinline__ uint32_t quad_sum(uint32_t v) {
    static_assert(LPS == 2 || LPS == 4 || LPS == 8 || LPS == 16, "2, 4, 8 or 16 lanes per symbol");
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);       // quad_perm [1,0,3,2]
    if constexpr (LPS >= 4)
        v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
    if constexpr (LPS >= 8)
        v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true);  // row_half_mirror
    if constexpr (LPS >= 16)
        v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, true);  // row_mirror
    return v;
}

// Bits 0, LPS, 2*LPS, ... of a wave-uniform mask packed into the low 64 / LPS bits (scalar unit).
template <int LPS>
__device__ __forceinline__ uint64_t compress_bits(uint64_t x) {
    if constexpr (LPS == 2) {
        x &= 0x5555555555555555ull;
        x = (x | (x >> 1)) & 0x3333333333333333ull;
        x = (x | (x >> 2)) & 0x0F0F0F0F0F0F0F0Full;
        x = (x | (x >> 4)) & 0x00FF00FF00FF00FFull;
        x = (x | (x >> 8)) & 0x0000FFFF0000FFFFull;
        x = (x | (x >> 16)) & 0x00000000FFFFFFFFull;
    } else if constexpr (LPS == 4) {
        x &= 0x1111111111111111ull;
        x = (x | (x >> 3)) & 0x0303030303030303ull;
        x = (x | (x >> 6)) & 0x000F000F000F000Full;
        x = (x | (x >> 12)) & 0x000000FF000000FFull;
        x = (x | (x >> 24)) & 0x000000000000FFFFull;
    } else if constexpr (LPS == 8) {
        x &= 0x0101010101010101ull;
        x = (x | (x >> 7)) & 0x0003000300030003ull;
        x = (x | (x >> 14)) & 0x0000000F0000000Full;
        x = (x | (x >> 28)) & 0x00000000000000FFull;
    } else {
        static_assert(LPS == 16, "2, 4, 8 or 16 lanes per symbol");
        x &= 0x0001000100010001ull;
        x = (x | (x >> 15)) & 0x0000000300000003ull;
        x = (x | (x >> 30)) & 0x000000000000000Full;
    }
    return x;
}

// The "loud enough" ballot of a pass with LPS lanes per symbol -> one bit per symbol (bit 0 / the LAST lane of each
// group carries the symbol's sum).  All data passes of a stream but its last have no quiet symbol: the 10 - 18 scalar
// instructions of the compaction are only spent when one of the deciding lanes is quiet (r5; all ones = nothing to
// stop, see rxd_stop).
template <int LPS, bool LAST, class CompressFn>
__device__ __forceinline__ uint64_t amp_ok_word(uint64_t loud, CompressFn&& compress) {
    if constexpr (LPS == 1) {
        return loud;
    } else {
        uint64_t sel = 0;
#pragma unroll
        for (int g = 0; g < 64; g += LPS) sel |= 1ull << (g + (LAST ? LPS - 1 : 0));
        if ((~loud & sel) == 0) return ~0ull;
        return compress(loud);
    }
}

}  // namespace afsk
