// afsk_demod_rounds_gp.h -- part of the single-pass demodulator (afsk_demod_fast.h includes the parts in order; see its header
// comment for the overall design): round loop of the general pieces: every other bit_frames a Receiver can have, as a compile-time value.
#pragma once

namespace afsk {

// ---- general pieces: any bit_frames as a COMPILE-TIME value (uniform kernels) ------------------------
// The remaining rates a Receiver can be built for (48000 / baud a divisor of 48000 and a multiple of 4:
// bit_frames 128, 192, 200, 300, 384, 400, 500, 600, 640, 800, 960, 1000, 1200, 1500, 1600, 1920, 2000
// = 375 ... 24 baud) have symbols that neither tile a round of chunks nor split into 2^k equal pieces of
// whole dwords inside one quarter (quarter lengths like 75 or 125 samples).  Here a symbol is split over
// LPS = 4 ... 64 lanes at DWORD granularity: quarter k of the symbol (template constant over it: mark
// hi,lo,hi,lo ref:80-85, space hi,hi,lo,lo ref:68-77) owns the dwords whose first sample lies in it, and
// its LPS/4 lanes share them as evenly as whole dwords allow -- every lane gets NB + 1 or NB + 2
// consecutive dwords.  Only the LAST dword of a lane can straddle into the next quarter (odd quarter
// length), so a lane runs NB dwords against its constant template (ONE v_sad_u16 against "hi" per dword
// serves both correlators: SAD against lo = 65535 * n - SAD against hi) and two tail slots with per-lane
// template dwords (the second one masked off for lanes with NB + 1 dwords).  All 64 lanes work for every
// bit_frames; rounds are 3.8 - 8 KiB of whole symbols with the watermark refill, linear reads into the
// mirror behind the ring, dword reads at the aligned address at or below a piece (shifted by 2 bytes in registers for an
// odd clock index; everything else of a clock index is taken out by FastRing::rebase).
template <int BF>
struct GpGeom {
    static constexpr int Q = BF / 4, D = BF / 2;
    static constexpr int pick_lps() {
        int l = 4;
        while (l < 64 && (64 / l) * 2 * BF > 8192) l *= 2;   // (7680: bit_frames 500 / 1000 / 2000 take half the pieces, 3 - 7 % slower at 4096 streams)
        return l;
    }
    static constexpr int LPS = pick_lps();                        // lanes per symbol
    static constexpr int LPQ = LPS / 4;                           // lanes per quarter symbol
    static constexpr int SPP = 64 / LPS;                          // symbols per round = per rxd pass
    static constexpr int RBYTES = SPP * 2 * BF;                   // bytes per round
    static constexpr int qs(int k) { return (k * Q + 1) / 2; }    // first dword owned by quarter k
    static constexpr int piece(int k, int j) {                    // dwords of lane j of quarter k
        return ((j + 1) * (qs(k + 1) - qs(k))) / LPQ - (j * (qs(k + 1) - qs(k))) / LPQ;
    }
    static constexpr int min_piece() {
        int m = 1 << 30;
        for (int k = 0; k < 4; k++) for (int j = 0; j < LPQ; j++) m = piece(k, j) < m ? piece(k, j) : m;
        return m;
    }
    static constexpr int max_piece() {
        int m = 0;
        for (int k = 0; k < 4; k++) for (int j = 0; j < LPQ; j++) m = piece(k, j) > m ? piece(k, j) : m;
        return m;
    }
    static constexpr int NB = min_piece() - 1;                    // dwords every lane runs against its constant template
    // largest power of two (bytes, at most 16) that divides the offset of every lane piece inside a round
    static constexpr int piece_align() {
        int a = 16;
        while (a > 4 && (2 * BF) % a != 0) a /= 2;
        for (int k = 0; k < 4; k++)
            for (int j = 0; j < LPQ; j++) {
                const int d0 = qs(k) + (j * (qs(k + 1) - qs(k))) / LPQ;
                while (a > 4 && (4 * d0) % a != 0) a /= 2;
            }
        return a;
    }
    static constexpr int PALIGN = piece_align();
    static constexpr bool valid = BF % 4 == 0 && BF >= 64 && 2 * BF < kSync && NB >= 1 && max_piece() <= NB + 2 &&
                                  4 * (NB + 3) <= kMirrorBytes && RBYTES + 4 + 1023 < kRingBytes;
};

// sum over the LPS lanes of an aligned group; the result is valid in the LAST lane of the group (for LPS
// <= 16 in every lane: DPP inside a row; 32 / 64 lanes add the row totals with row_bcast:15 / :31)
template <int LPS>
__device__ __forceinline__ uint32_t group_sum_last(uint32_t v) {
    if constexpr (LPS <= 16) {
        return quad_sum<LPS>(v);
    } else {
        v = quad_sum<16>(v);
        v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);       // row_bcast:15 -> rows 1, 3
        if constexpr (LPS == 64)
            v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2, 3
        return v;
    }
}

// bit (g * LPS + LPS - 1) of a wave-uniform mask -> bit g (the last lane of every group)
template <int LPS>
__device__ __forceinline__ uint64_t compress_bits_last(uint64_t x) {
    if constexpr (LPS == 64) return x >> 63;
    else if constexpr (LPS == 32) return ((x >> 31) & 1ull) | ((x >> 62) & 2ull);
    else return compress_bits<LPS>(x >> (LPS - 1));
}

// ODD: byte0 is 2 bytes past a multiple of 16 (an odd clock index after FastRing::rebase), else a multiple of 16.
template <int BF, int FLAGS, bool ODD, bool HINTED>
__device__ __forceinline__ void gp_rounds(FastRing& fr, int byte0, int32_t K, int32_t NR,
                                          uint32_t amp_thr, RxDeferred& rd,
                                          unsigned long long* words, uint8_t* out_row,
                                          int out_stride, int32_t* margins, int32_t mstride) {
    using G = GpGeom<BF>;
    static_assert(G::valid, "no general-piece geometry for this bit_frames");
    constexpr int Q = G::Q, LPS = G::LPS, LPQ = G::LPQ, SPP = G::SPP, RBYTES = G::RBYTES, NB = G::NB;
    constexpr uint32_t FULL = 65535u;
    const int lane = fr.lane;
    const int part = lane & (LPS - 1), sym = lane / LPS;
    // this lane's piece of every symbol it works on: dwords [d0, d0 + n) of quarter k
    const int k = part / LPQ, j = part % LPQ;
    const int q0 = (k * Q + 1) >> 1, q1 = ((k + 1) * Q + 1) >> 1;
    const int d0 = q0 + (j * (q1 - q0)) / LPQ, d1 = q0 + ((j + 1) * (q1 - q0)) / LPQ;
    const bool two = (d1 - d0) == NB + 2;                         // NB + 2 dwords (else NB + 1)
    const bool mark_hi = (k & 1) == 0, space_hi = k < 2;          // quarter k: mark hi,lo,hi,lo / space hi,hi,lo,lo
    const uint32_t cm = mark_hi ? 0xFFFFFFFFu : 0u, cs = space_hi ? 0xFFFFFFFFu : 0u;
    // the last dword of the piece: its second sample may already belong to the next quarter
    const int kl = (2 * d1 - 1) / Q;
    const uint32_t lm = (cm & 0xFFFFu) | (((kl & 1) == 0 ? 0xFFFFu : 0u) << 16);
    const uint32_t ls = (cs & 0xFFFFu) | ((kl < 2 ? 0xFFFFu : 0u) << 16);
    const uint32_t tmA = two ? cm : lm, tsA = two ? cs : ls;      // tail slot A = dword NB of the piece
    const int piece_byte = sym * 2 * BF + 4 * d0;
    int pos = byte0;                                              // stream byte of the round's first sample
    for (int r = 0; r < NR; r++, pos += RBYTES) {
        const int last = pos + RBYTES + 3;                        // tail slot B of the last lane reaches one dword further
        bool partial;                                             // (a partial round: see FastRing::holding_wait)
        RxDeferred saved;
        const int32_t Kr = fr.template wait_round<(FLAGS & 4) ? 0 : 2, RBYTES + 4, HINTED>(pos, K, r * SPP, byte0, 2 * BF, partial);
        if (HINTED && partial) saved = rd;
        const int32_t mlim = Kr < mstride ? Kr : mstride;         // soft output rows hold symbols [0, mlim)
        if constexpr (HINTED) fr.template eval_probes<fine_probes(RBYTES)>(last >> 10, amp_thr / (uint32_t)BF, byte0, 4, 2 * BF);
        const int rb = pos & (kRingBytes - 1);
        if (rb + RBYTES + 4 > kRingBytes) {                       // a piece runs past the ring end: refresh the mirror
            if (lane < kMirrorBytes / 16)
                *reinterpret_cast<u32x4*>(fr.ring + kRingBytes + 16 * lane) =
                    *reinterpret_cast<const u32x4*>(fr.ring + 16 * lane);
            wave_lds_sync();
        }
        // The piece is read from the dword-aligned address at or below its first byte (an odd clock index puts
        // it 2 bytes into a dword) and shifted in registers.  The reads are typed by what is KNOWN about that
        // address, because the compiler merges adjacent dword reads into 8- and 16-byte reads and the hardware
        // executes those several times slower at addresses that are not that aligned (5 us per 4096 streams at
        // 240 / 160 / 120 / 80 baud, 15 % at 32768 x 160 baud): geometries whose pieces all start on 16- (8-)
        // byte multiples of the round read 16 (8) bytes at a time when the clock index allows it (bit_frames
        // 192, 384, 640 ...: 6 % faster than dword pairs); everything else reads dword pairs (ds_read2_b32
        // needs 4-byte alignment only).
        constexpr int sh = ODD ? 2 : 0;                               // 2 for an odd clock index
        const uint8_t* src = fr.ring + (((rb + piece_byte) & (kRingBytes - 1)) - sh);
        constexpr int PALIGN = G::PALIGN;
        constexpr int NW = NB + 3;                                    // one dword more for the shifted form
        uint32_t W[NW];
        if constexpr (PALIGN >= 8) {                                  // (src = the 16-byte-aligned ring byte of the round + a multiple of PALIGN)
            constexpr int VW = PALIGN / 4;                            // dwords per read
            typedef uint32_t uvec __attribute__((ext_vector_type(VW), aligned(PALIGN)));
#pragma unroll
            for (int v = 0; v < NW / VW; v++) {
                const uvec t = *reinterpret_cast<const uvec*>(src + PALIGN * v);
#pragma unroll
                for (int u = 0; u < VW; u++) W[VW * v + u] = t[u];
            }
#pragma unroll
            for (int d = (NW / VW) * VW; d < NW; d++) W[d] = *reinterpret_cast<const uint32_t*>(src + 4 * d);
        } else {
#pragma unroll
            for (int d = 0; d < NW; d++) W[d] = *reinterpret_cast<const uint32_t*>(src + 4 * d);
        }
        uint32_t x[NB + 2];
        if constexpr (sh == 0) {
#pragma unroll
            for (int d = 0; d < NB + 2; d++) x[d] = W[d];
        } else {
#pragma unroll
            for (int d = 0; d < NB + 2; d++) x[d] = __builtin_amdgcn_alignbyte(W[d + 1], W[d], 2);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // values are in registers: refill right away
        if (!(HINTED && partial)) fr.template refill_round<(FLAGS & 4) ? 0 : 2, RBYTES, HINTED>(pos);

        uint32_t h = 0;
#pragma unroll
        for (int d = 0; d < NB; d++)
            h = __builtin_amdgcn_sad_u16(limit_pair_biased(x[d]), 0xFFFFFFFFu, h);         // ref:344, 346-347
        const uint32_t la = limit_pair_biased(x[NB]), lb = limit_pair_biased(x[NB + 1]);
        uint32_t mark = mark_hi ? h : FULL * (2u * NB) - h;
        uint32_t space = space_hi ? h : FULL * (2u * NB) - h;
        mark = __builtin_amdgcn_sad_u16(la, tmA, mark);
        space = __builtin_amdgcn_sad_u16(la, tsA, space);
        const uint32_t mb = __builtin_amdgcn_sad_u16(lb, lm, 0u), sb = __builtin_amdgcn_sad_u16(lb, ls, 0u);
        mark += two ? mb : 0u;
        space += two ? sb : 0u;
        mark = group_sum_last<LPS>(mark);
        space = group_sum_last<LPS>(space);
        const int k0 = r * SPP;
        const uint32_t md = mark / (uint32_t)BF, sd = space / (uint32_t)BF;
        const bool bit = md < sd;                                                        // ref:348-351
        if (margins && part == LPS - 1 && k0 + sym < mlim) margins[k0 + sym] = (int32_t)sd - (int32_t)md;
        const int nv = (Kr - k0) < SPP ? (Kr - k0) : SPP;
        const uint64_t bmask = compress_bits_last<LPS>(__ballot(bit));
        // the squelch amplitude (ref:94-98, ref:375) is only formed in passes that hold data symbols -- the
        // reference does not evaluate it during training either (ref:361-366); r4: a quarter of the per-dword
        // VALU work of the training rounds
        rxd_pass<SPP>(rd, bmask, nv, k0, lane, words, out_row, out_stride, [&]() {
            uint32_t q = 0;                                   // quiet sums (see quiet_sad): the lanes of a symbol add up to 32768 BF - sum|x|
#pragma unroll
            for (int d = 0; d <= NB; d++) q = quiet_sad(x[d], q);
            const uint32_t qb = quiet_sad(x[NB + 1], 0u);
            q += two ? qb : 0u;
            const uint32_t qsum = group_sum_last<LPS>(q);
            return amp_ok_word<LPS, true>(__ballot(loud_enough(qsum, (uint32_t)BF, amp_thr)), [](uint64_t b) { return compress_bits_last<LPS>(b); });
        });
        if (rd.st.phase == 2) break;
        if (HINTED && partial) {           // no squelch stop among the symbols that were there: fetch the rest, run the round again
            rd = saved;
            fr.template fetch_through<(FLAGS & 4) ? 0 : 2>(last >> 10);
            r--; pos -= RBYTES;
        }
    }
}

}  // namespace afsk
