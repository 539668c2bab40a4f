// afsk_demod_rounds_wm.h -- part of the single-pass demodulator (afsk_demod_fast.h includes the parts in order; see its header
// comment for the overall design): round loop of bit_frames 60 / 96 / 100 / 120 / 128 / 240 / 320 / 480: rounds of any size, watermark refill.
#pragma once

namespace afsk {

// ---- bit_frames 60 / 96 / 100 / 120 (800 / 500 / 480 / 400 baud) on the single-pass ring ----
// Their symbols do not tile a round of whole 1 KiB chunks, so a round is 64 lane pieces of PB bytes
// (any multiple of 4) and the refill follows a consumed-byte WATERMARK: after the reads of a round
// every chunk that lies wholly below the next round's first byte is requested again, and the wait
// before a round is for the chunk holding its last byte (a wave-uniform count -> s_waitcnt through a
// scalar switch).  A lane reads its piece LINEARLY from (ring offset of its first byte) -- the one
// piece that straddles the ring end runs on into the 256-byte mirror of ring bytes 0..255 that the
// wave refreshes (one ds_read_b128 + ds_write_b128 by 16 lanes) in exactly the rounds that cross the
// end -- so adjacent 4- and 8-byte reads merge into ds_read2_b32 / ds_read2_b64.
//   bit_frames  60: one lane per symbol, 120-byte pieces (8-byte aligned), per-dword templates
//               96: two lanes per symbol, 96-byte pieces (six ds_read_b128), quarter sums
//              100: two lanes per symbol, 100-byte pieces (4-byte aligned), the quarter boundary
//                   falls inside a dword: mark SAD against a per-dword template + one "hi" SAD
//              120: two lanes per symbol, 120-byte pieces (8-byte aligned), quarter sums
template <int BF>
struct WmGeom {
    static constexpr bool valid = BF == 60 || BF == 96 || BF == 100 || BF == 120 || BF == 128 || BF == 240 || BF == 320 || BF == 480;
    // lanes per symbol: a whole symbol (60), half a symbol (96 / 100 / 120, and since r4 128 = 375 baud: 8 KiB
    // rounds of eight 16-byte reads per lane, 0.746 -> 0.792 of peak at 65536 streams against its general-piece
    // form with 4 KiB rounds), and for the long symbols of
    // 200 / 150 / 100 baud a piece that lies inside ONE quarter of the symbol (both templates constant
    // over it): 240 -> 4 x 60 samples, 320 -> 8 x 40, 480 -> 8 x 60
    static constexpr int LPS = BF >= 320 ? 8 : (BF >= 240 ? 4 : (BF >= 96 ? 2 : 1));
    static constexpr int PL = BF / LPS;                           // samples per lane piece
    static constexpr int PB = 2 * PL;                             // bytes per piece
    static constexpr int NO = PL / 2;                             // dwords per piece
    static constexpr int RW = PB % 16 == 0 ? 16 : (PB % 8 == 0 ? 8 : 4);   // natural alignment of a piece
    static constexpr int SPP = 64 / LPS;                          // symbols per round = per rxd pass
    static constexpr int RBYTES = 64 * PB;                        // bytes per round
    static_assert(!valid || (BF % 4 == 0 && PL % 2 == 0 && PB + 16 <= kMirrorBytes &&
                             RBYTES + 16 + 1023 < kRingBytes && (LPS == 1 || (BF / 2) % 2 == 0) &&
                             (LPS < 4 || (BF / 4) % PL == 0)),
                  "round geometry");
};

// ODD: byte0 is 2 bytes past a multiple of 16 (an odd clock index after FastRing::rebase): the reads start at the
// aligned address below and take one unit more, shifted down in registers; otherwise byte0 is a multiple of 16.
template <int BF, int FLAGS, bool ODD, bool HINTED>
__device__ __forceinline__ void wm_rounds(FastRing& fr, int byte0, int32_t K, int32_t NR,
                                          uint32_t amp_thr, RxDeferred& rd,
                                          unsigned long long* words, uint8_t* out_row,
                                          int out_stride, int32_t* margins, int32_t mstride) {
    using G = WmGeom<BF>;
    constexpr int LPS = G::LPS, PL = G::PL, PB = G::PB, NO = G::NO, RW = G::RW, SPP = G::SPP, RBYTES = G::RBYTES;
    constexpr int Q = BF / 4, H = BF / 2;
    constexpr uint32_t FULL = 65535u;
    constexpr bool ALIGNED = !ODD;
    constexpr int EXTRA = ALIGNED ? 0 : RW;                       // the re-aligning path reads one unit more
    constexpr int NW = NO + EXTRA / 4;                            // dwords a lane reads
    typedef u32x4 u32x4_a16 __attribute__((aligned(16)));
    typedef u32x2 u32x2_a8 __attribute__((aligned(8)));
    const int lane = fr.lane;
    const int part = lane & (LPS - 1);
    // 128-byte pieces (bit_frames 128): sixteen lanes of a ds_read_b128 group, 128 bytes apart, would meet on two
    // bank quads -- an 8-way conflict on every read (r5 PMC: 79 % of the LDS cycles of this kernel).  A lane piece
    // is two quarters of four 16-byte chunks, and inside a quarter the order of the chunks does not matter (one
    // template, one sum): read j takes chunk (j + r) & 3 of quarter (j >> 2) ^ sw, with r = lane bits 1-2 and
    // sw = lane bit 3 -- the 16 lanes of a group then touch 16 different bank quads -- and the two quarter sums
    // are exchanged in the lanes with sw set.
    constexpr bool SWZ = ALIGNED && BF == 128;
    int swz_off[SWZ ? 8 : 1];
    const bool swz_sw = SWZ && ((lane >> 3) & 1);
    if constexpr (SWZ) {
#pragma unroll
        for (int j = 0; j < 8; j++)
            swz_off[j] = 16 * (((j & 3) + ((lane >> 1) & 3)) & 3) + 64 * ((j >> 2) ^ ((lane >> 3) & 1));
    }
    int pos = byte0 & ~(RW - 1);                                  // stream byte where this round's reads start
    for (int r = 0; r < NR; r++, pos += RBYTES) {
        const int last = pos + RBYTES + EXTRA - 1;                // last stream byte this round reads
        bool partial;                                             // (a partial round: see FastRing::holding_wait)
        RxDeferred saved;
        const int32_t Kr = fr.template wait_round<(FLAGS & 4) ? 0 : 2, RBYTES + EXTRA, HINTED>(pos, K, r * SPP, byte0, 2 * BF, partial);
        if (HINTED && partial) saved = rd;
        const int32_t mlim = Kr < mstride ? Kr : mstride;         // soft output rows hold symbols [0, mlim)
        if constexpr (HINTED) fr.template eval_probes<fine_probes(RBYTES)>(last >> 10, amp_thr / (uint32_t)BF, byte0 & ~(RW - 1), EXTRA, 2 * BF);
        const int rb = pos & (kRingBytes - 1);                    // wave-uniform
        if (rb + RBYTES + EXTRA > kRingBytes) {                   // a piece runs past the ring end: refresh the mirror
            if (lane < kMirrorBytes / 16)
                *reinterpret_cast<u32x4*>(fr.ring + kRingBytes + 16 * lane) =
                    *reinterpret_cast<const u32x4*>(fr.ring + 16 * lane);
            wave_lds_sync();
        }
        const uint8_t* src = fr.ring + ((rb + PB * lane) & (kRingBytes - 1));
        uint32_t W[NW];
#pragma unroll
        for (int j = 0; j < NW * 4 / RW; j++) {
            if constexpr (SWZ) {
                const u32x4 t4 = *reinterpret_cast<const u32x4_a16*>(src + swz_off[j]);
                W[4 * j] = t4[0]; W[4 * j + 1] = t4[1]; W[4 * j + 2] = t4[2]; W[4 * j + 3] = t4[3];
            } else if constexpr (RW == 16) {
                const u32x4 t4 = *reinterpret_cast<const u32x4_a16*>(src + 16 * j);
                W[4 * j] = t4[0]; W[4 * j + 1] = t4[1]; W[4 * j + 2] = t4[2]; W[4 * j + 3] = t4[3];
            } else if constexpr (RW == 8) {
                const u32x2 t2 = *reinterpret_cast<const u32x2_a8*>(src + 8 * j);
                W[2 * j] = t2[0]; W[2 * j + 1] = t2[1];
            } else {
                W[j] = *reinterpret_cast<const uint32_t*>(src + 4 * j);
            }
        }
        uint32_t x[NO];
        if constexpr (ALIGNED) {
#pragma unroll
            for (int d = 0; d < NO; d++) x[d] = W[d];
        } else {
            realign_n<2, NW, NO>(W, x);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // values are in registers: refill right away
        // every chunk wholly below the next round's first byte is free
        if (!(HINTED && partial)) fr.template refill_round<(FLAGS & 4) ? 0 : 2, RBYTES, HINTED>(pos);

        const int k0 = r * SPP;
        uint32_t mark = 0, space = 0;
        if constexpr (LPS == 1) {                                 // whole symbol in the lane: per-dword templates
#pragma unroll
            for (int d = 0; d < NO; d++) {
                const uint32_t lim = limit_pair_biased(x[d]);                              // ref:344
                const uint32_t tm = mark_half(2 * d, Q) | (mark_half(2 * d + 1, Q) << 16);
                const uint32_t ts = space_half(2 * d, H) | (space_half(2 * d + 1, H) << 16);
                mark = __builtin_amdgcn_sad_u16(lim, tm, mark);                          // ref:346
                space = __builtin_amdgcn_sad_u16(lim, ts, space);                        // ref:347
            }
        } else if constexpr (LPS >= 4) {
            // the piece lies inside quarter `part / (LPS / 4)` of the symbol: mark template hi,lo,hi,lo over
            // the quarters (ref:80-85), space template hi,hi,lo,lo (ref:68-77), both constant over the piece
            uint32_t h = 0;
#pragma unroll
            for (int d = 0; d < NO; d++) h = __builtin_amdgcn_sad_u16(limit_pair_biased(x[d]), 0xFFFFFFFFu, h);
            const int quarter = part / (LPS / 4);
            mark = (quarter & 1) ? FULL * PL - h : h;
            space = quarter < 2 ? h : FULL * PL - h;
        } else if constexpr (Q % 2 == 0) {
            // half a symbol in the lane = quarters (hi, lo) of the mark tone (ref:80-85), all hi (part 0)
            // or all lo (part 1) of the space tone (ref:68-77); SAD against lo = 65535 * n - SAD against hi
            uint32_t ha = 0, hb = 0;
#pragma unroll
            for (int d = 0; d < Q / 2; d++) ha = __builtin_amdgcn_sad_u16(limit_pair_biased(x[d]), 0xFFFFFFFFu, ha);
#pragma unroll
            for (int d = Q / 2; d < Q; d++) hb = __builtin_amdgcn_sad_u16(limit_pair_biased(x[d]), 0xFFFFFFFFu, hb);
            if constexpr (SWZ) {               // lanes that read their second quarter first
                const uint32_t t = ha;
                ha = swz_sw ? hb : ha;
                hb = swz_sw ? t : hb;
            }
            mark = ha + (FULL * Q - hb);
            space = part == 0 ? ha + hb : 2u * FULL * Q - ha - hb;
        } else {
            // odd quarter length: sample Q - 1 | Q share a dword, so the mark SAD uses per-dword
            // templates; the space SAD follows from the SAD against "hi" of the whole piece
            uint32_t mk = 0, th = 0;
#pragma unroll
            for (int d = 0; d < NO; d++) {
                const uint32_t lim = limit_pair_biased(x[d]);
                const uint32_t tm = mark_half(2 * d, Q) | (mark_half(2 * d + 1, Q) << 16);   // phases < H: hi Q, lo Q
                mk = __builtin_amdgcn_sad_u16(lim, tm, mk);
                th = __builtin_amdgcn_sad_u16(lim, 0xFFFFFFFFu, th);
            }
            mark = mk;
            space = part == 0 ? th : FULL * PL - th;
        }
        if constexpr (LPS >= 2) {
            mark = quad_sum<LPS>(mark);
            space = quad_sum<LPS>(space);
        }
        const uint32_t md = mark / (uint32_t)BF, sd = space / (uint32_t)BF;
        const bool bit = md < sd;                                                        // ref:348-351
        if (margins && part == 0 && k0 + lane / LPS < mlim) margins[k0 + lane / LPS] = (int32_t)sd - (int32_t)md;
        const int nv = (Kr - k0) < SPP ? (Kr - k0) : SPP;
        uint64_t bmask = __ballot(bit);
        if constexpr (LPS >= 2) bmask = compress_bits<LPS>(bmask);
        rxd_pass<SPP>(rd, bmask, nv, k0, lane, words, out_row, out_stride, [&]() {
            uint32_t q = 0;
#pragma unroll
            for (int d = 0; d < NO; d++) q = quiet_sad(x[d], q);                                      // ref:94-98
            if constexpr (LPS >= 2) q = quad_sum<LPS>(q);
            return amp_ok_word<LPS, false>(__ballot(loud_enough(q, (uint32_t)BF, amp_thr)), [](uint64_t b) { if constexpr (LPS >= 2) return compress_bits<LPS>(b); else return b; });
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
