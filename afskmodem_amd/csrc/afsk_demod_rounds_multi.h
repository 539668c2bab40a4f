// afsk_demod_rounds_multi.h -- part of the single-pass demodulator (afsk_demod_fast.h includes the parts in order; see its header
// comment for the overall design): round loop of bit_frames 4 / 8 / 12 / 16 / 24 / 32 / 48 / 64 (12000 ... 750 baud): several whole symbols per lane.
#pragma once

namespace afsk {

// ---- other baud rates on the single-pass ring: several whole symbols per lane ------------
// bit_frames 4 / 8 / 12 / 16 / 24 / 32 / 48 / 64 (12000 ... 750 baud).  A round is R chunks =
// 64 * SPL symbols; lane l takes symbols l, l + 64, ... (SPL pieces of 2*BF bytes, read with
// 16-byte loads when BF % 8 == 0, 8-byte loads otherwise), so every 64-symbol slice of the round is
// one plain ballot -- the 2400-baud scheme with other sizes.  (60, 96, 100 and 120 have no round of
// whole chunks that leaves enough of the ring in flight: wm_rounds below.)
template <int BF>
struct MultiGeom {
    static constexpr bool valid = BF == 4 || BF == 8 || BF == 12 || BF == 16 || BF == 24 || BF == 32 ||
                                  BF == 48 || BF == 64;
    // chunks per round (overridable per value for A/B builds: -DAFSK_R16=8 ...)
#ifndef AFSK_R4
#define AFSK_R4 5
#endif
#ifndef AFSK_R8
#define AFSK_R8 5
#endif
#ifndef AFSK_R12
#define AFSK_R12 6
#endif
#ifndef AFSK_R16
#define AFSK_R16 8      // r5: 8 KiB rounds of four slices, -2.2 % at 65536 streams, -2.8 % at 4096 (profiles/r5_exp10_chunks_per_round.txt;
#endif                  // bit_frames 12: 9 against 6 neutral; 24: 9 costs 5 %; 8: 6 / 8 cost 5 % / 2 %)
#ifndef AFSK_R24
#define AFSK_R24 6
#endif
    static constexpr int R = BF == 4 ? AFSK_R4 : (BF == 8 ? AFSK_R8 : (BF == 12 ? AFSK_R12 : (BF == 16 ? AFSK_R16 : (BF == 24 ? AFSK_R24 :
                             (BF == 32 ? 4 : (BF == 64 ? 8 : 6))))));
    static constexpr int SPL = 8 * R / BF;                     // symbols per lane per round
    static constexpr int PB = 2 * BF;                          // bytes per symbol
    static constexpr int RW = BF % 8 == 0 ? 16 : 8;            // bytes per LDS read
    static constexpr int NO = BF / 2;                          // dwords per symbol
    static constexpr int SPR = 64 * SPL;                       // symbols per round
    static_assert(!valid || (SPL * BF == 8 * R && PB % RW == 0 && R + 1 < kRingChunks), "round geometry");
};

// ODD: byte0 is 2 bytes past a multiple of 16 (an odd clock index after FastRing::rebase): one read unit more per
// piece, shifted down in registers; otherwise byte0 is a multiple of 16.
template <int BF, int FLAGS, bool ODD, bool HINTED>
__device__ __forceinline__ void multi_rounds(FastRing& fr, int byte0, int32_t K, int32_t NR,
                                             uint32_t amp_thr, RxDeferred& rd,
                                             unsigned long long* words, uint8_t* out_row,
                                             int out_stride, int32_t* margins, int32_t mstride) {
    using MG = MultiGeom<BF>;
    constexpr int R = MG::R, SPL = MG::SPL, PB = MG::PB, RW = MG::RW, NO = MG::NO, SPR = MG::SPR;
    constexpr bool ALIGNED = !ODD;
    constexpr int Q = BF / 4, H = BF / 2;
    constexpr uint32_t FULL = 65535u;
    constexpr int NR_READS = PB / RW;                          // reads per piece when aligned
    constexpr int DW = RW / 4;                                 // dwords per read
    const int lane = fr.lane;
    // (bit_frames 16 / 32 / 64: the lanes of a ds_read_b128 group, 32 / 64 / 128 bytes apart, collide on bank quads -- 2-,
    // 4-, 8-way.  A conflict-free read order was measured for bit_frames 16 (second half first in half the lanes, space
    // correlator sign-flipped there): +-0, profiles/r5_exp12_swap16.txt -- LDS instructions are 1 - 2 % of what a wave issues.)
    for (int r = 0; r < NR; r++) {
        // bytes [byte0 + 1024 R r, +1024 R) must have landed: at most R + 1 chunks from the oldest
        // resident one; chunks through B_r + 15 are issued, so the 15 - R youngest may be in flight
        int32_t Kr = K;                    // symbols this round may use (fewer: a partial round, see holding_wait)
        bool partial = false;
        RxDeferred saved;
        const int last = byte0 + 1024 * R * (r + 1) - 1 + (ALIGNED ? 0 : RW);            // last byte read
        if (HINTED && fr.hint_holding()) { // the tail hint has stopped the fixed R-chunks-per-round schedule
            Kr = fr.template holding_wait<(FLAGS & 4) ? 0 : 2>(last, K, r * SPR, byte0, PB, partial);
            if (partial) saved = rd;
        } else {
            fr.template wait_fixed<kRingChunks - 1 - R>(((byte0 + 1024 * R * r) >> 10) + R);
            if constexpr (HINTED) fr.template eval_probes<fine_probes(1024 * R)>(((byte0 + 1024 * R * r) >> 10) + R, amp_thr / (uint32_t)BF, byte0, ALIGNED ? 0 : RW, PB);
        }
        uint32_t x[SPL * NO];
        const int rb = (byte0 + 1024 * R * r) & (kRingBytes - 1);      // wave-uniform
        // a round that does not cross the ring end (two of three) reads at constant offsets from ONE lane address
        // (r5: the masked form costs three VALU instructions per read for the wrap that mostly does not happen)
        const bool nowrap = ALIGNED && rb + 1024 * R <= kRingBytes;    // wave-uniform
        auto read_piece = [&](const uint8_t* p, int piece, int j) {
            if constexpr (RW == 16) {
                const u32x4 t4 = *reinterpret_cast<const u32x4*>(p);
                x[NO * piece + 4 * j] = t4[0]; x[NO * piece + 4 * j + 1] = t4[1];
                x[NO * piece + 4 * j + 2] = t4[2]; x[NO * piece + 4 * j + 3] = t4[3];
            } else {
                const u32x2 t2 = *reinterpret_cast<const u32x2*>(p);
                x[NO * piece + 2 * j] = t2[0]; x[NO * piece + 2 * j + 1] = t2[1];
            }
        };
        if (nowrap) {
            const uint8_t* src = fr.ring + rb + PB * lane;
#pragma unroll
            for (int piece = 0; piece < SPL; piece++)
#pragma unroll
                for (int j = 0; j < NR_READS; j++) read_piece(src + 64 * PB * piece + RW * j, piece, j);
            asm volatile("" ::: "memory");                             // (keeps the compiler from merging the two forms into selects)
        } else {
#pragma unroll
        for (int piece = 0; piece < SPL; piece++) {
            const int pb = rb + 64 * PB * piece + PB * lane;
            if constexpr (ALIGNED) {
#pragma unroll
                for (int j = 0; j < NR_READS; j++)
                    read_piece(fr.ring + ((pb + RW * j) & (kRingBytes - 1)), piece, j);
            } else {
                const int ab = pb & ~(RW - 1);
                uint32_t W[NO + DW];
#pragma unroll
                for (int j = 0; j < NR_READS + 1; j++) {
                    const uint8_t* p = fr.ring + ((ab + RW * j) & (kRingBytes - 1));
                    if constexpr (RW == 16) {
                        const u32x4 t4 = *reinterpret_cast<const u32x4*>(p);
                        W[4 * j] = t4[0]; W[4 * j + 1] = t4[1]; W[4 * j + 2] = t4[2]; W[4 * j + 3] = t4[3];
                    } else {
                        const u32x2 t2 = *reinterpret_cast<const u32x2*>(p);
                        W[2 * j] = t2[0]; W[2 * j + 1] = t2[1];
                    }
                }
                uint32_t y[NO];
                realign_n<2, NO + DW, NO>(W, y);
#pragma unroll
                for (int d = 0; d < NO; d++) x[NO * piece + d] = y[d];
            }
        }
        }
        const int k0 = r * SPR;
        const int32_t mlim = Kr < mstride ? Kr : mstride;       // soft output rows hold symbols [0, mlim)
        uint64_t B[SPL];
        int32_t mg[SPL];                                                       // space_diff - mark_diff per slice (soft output)
        uint32_t l12[BF == 4 ? SPL : 1];                                       // bit_frames 4: the limited (sample 1, sample 2) pairs
        auto decide = [&](auto pc) {
            constexpr int piece = decltype(pc)::value;
            if constexpr (BF == 4) {
                // One sample per quarter.  With the limited samples L0..L3 (biased levels 0 / 0x8000 / 0xFFFF)
                // mark = (65535 - L0) + L1 + (65535 - L2) + L3 and space = (65535 - L0) + (65535 - L1) + L2 + L3
                // (ref:80-85, 68-77, 346-347), so mark - space = 2 (L1 - L2): equal levels tie (bit 0, ref:350),
                // different levels differ by at least 32767, far more than the truncation of the two means can
                // hide -- the decision int(mark / 4) < int(space / 4) IS L1 < L2 (exhaustive check:
                // tests/test_kernel_math.py).  The two quotients themselves are only needed for the margins.
                // Only samples 1 and 2 decide: one dword holding both goes through ONE limiter.  (The margins,
                // which need all four samples, are formed after the loop; the ten compares follow it too.)
                l12[piece] = limit_pair_biased(__builtin_amdgcn_alignbit(x[NO * piece + 1], x[NO * piece], 16));   // (sample 1, sample 2)
            } else {
                uint32_t mark = 0, space = 0;
                if constexpr (BF % 8 == 0) {
                    // quarters are whole dwords: SAD against "hi" per quarter gives both correlators
                    uint32_t hq[4] = {0, 0, 0, 0};
#pragma unroll
                    for (int d = 0; d < NO; d++)
                        hq[d / (Q / 2)] = __builtin_amdgcn_sad_u16(limit_pair_biased(x[NO * piece + d]), 0xFFFFFFFFu,
                                                                  hq[d / (Q / 2)]);
                    const uint32_t u = 2u * FULL * Q + hq[0] - hq[3], dd = hq[2] - hq[1];   // (modulo 2^32, like the sums)
                    mark = u + dd;
                    space = u - dd;
                } else {
#pragma unroll
                    for (int d = 0; d < NO; d++) {
                        const uint32_t lim = limit_pair_biased(x[NO * piece + d]);
                        const uint32_t tm = mark_half(2 * d, Q) | (mark_half(2 * d + 1, Q) << 16);
                        const uint32_t ts = space_half(2 * d, H) | (space_half(2 * d + 1, H) << 16);
                        mark = __builtin_amdgcn_sad_u16(lim, tm, mark);
                        space = __builtin_amdgcn_sad_u16(lim, ts, space);
                    }
                }
                const uint32_t md = mark / (uint32_t)BF, sd = space / (uint32_t)BF;
                mg[piece] = (int32_t)sd - (int32_t)md;
                if constexpr ((BF & (BF - 1)) == 0)
                    // floor(mark / BF) < floor(space / BF)  <=>  mark < (space with its low log2(BF) bits cleared): one
                    // v_and + v_cmp instead of two shifts + v_cmp (the quotients above are only formed for the margins)
                    B[piece] = __ballot(mark < (space & ~(uint32_t)(BF - 1)));     // ref:348-351
                else
                    B[piece] = __ballot(md < sd);                                  // ref:348-351
            }
        };
        // (forming the first slices' decisions BEFORE the refill, while the later reads are in flight, was measured:
        // +-1 %, profiles/r5_exp9_early.txt)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // values are in x: refill
        if (HINTED && partial) {
            // (no refill: the round may have to run again on the same ring contents)
        } else if (HINTED && fr.hint_takes_over(R)) {
            fr.template top_up<(FLAGS & 4) ? 0 : 2>(((byte0 + 1024 * R * (r + 1)) >> 10) + kRingChunks);
        } else {
            fr.template issue_run<(FLAGS & 4) ? 0 : 2, R>(fr.next);
            fr.next += R;
        }
        static_for<0, SPL>(decide);
        if constexpr (BF == 4) {
            static_assert(BF != 4 || SPL == 10, "ten slices per round");
            // L1 < L2 (ref:348-351) as ONE 16-bit compare of the two halves of a register (SDWA operand selects): the
            // compiler forms the same test from a mask, a shift and a 32-bit compare.  One asm statement, closed by
            // s_nop 1: the ballots are SGPRs written by VALU, and whatever VALU instruction reads them next (the
            // spreading of the words over the lanes) must be two wait states behind (see spread_words).
#define AFSK_C(i) "v_cmp_lt_u16_sdwa %" #i ", %1" #i ", %1" #i " src0_sel:WORD_0 src1_sel:WORD_1\n\t"
            asm volatile(AFSK_C(0) AFSK_C(1) AFSK_C(2) AFSK_C(3) AFSK_C(4) AFSK_C(5) AFSK_C(6) AFSK_C(7) AFSK_C(8) AFSK_C(9) "s_nop 1"
                         : "=s"(B[0]), "=s"(B[1]), "=s"(B[2]), "=s"(B[3]), "=s"(B[4]), "=s"(B[5]), "=s"(B[6]), "=s"(B[7]), "=s"(B[8]), "=s"(B[9])
                         : "v"(l12[0]), "v"(l12[1]), "v"(l12[2]), "v"(l12[3]), "v"(l12[4]), "v"(l12[5]), "v"(l12[6]), "v"(l12[7]), "v"(l12[8]), "v"(l12[9]));
#undef AFSK_C
        }
        if constexpr (BF != 4) {
            if (margins) {                 // soft output, ONE test per round (r4: it sat inside the slice loop)
#pragma unroll
                for (int piece = 0; piece < SPL; piece++)
                    if (k0 + 64 * piece + lane < mlim) margins[k0 + 64 * piece + lane] = mg[piece];
            }
        }
        if constexpr (BF == 4) {
            if (margins) {                                                     // soft output: the two quotients (ref:346-349)
#pragma unroll
                for (int piece = 0; piece < SPL; piece++) {
                    const int kk = k0 + 64 * piece;
                    const uint32_t l0 = limit_pair_biased(x[NO * piece]), l1 = limit_pair_biased(x[NO * piece + 1]);
                    const uint32_t mk = __builtin_amdgcn_sad_u16(l1, 0x0000FFFFu, __builtin_amdgcn_sad_u16(l0, 0x0000FFFFu, 0u));
                    const uint32_t sp = __builtin_amdgcn_sad_u16(l1, 0x00000000u, __builtin_amdgcn_sad_u16(l0, 0xFFFFFFFFu, 0u));
                    if (kk + lane < mlim) margins[kk + lane] = (int32_t)(sp / 4u) - (int32_t)(mk / 4u);
                }
            }
        }
        auto amp_word = [&](int piece) {                                       // ref:94-98, ref:375
            uint32_t q = 0;
#pragma unroll
            for (int d = 0; d < NO; d++) q = quiet_sad(x[NO * piece + d], q);
            return __ballot(loud_enough(q, (uint32_t)BF, amp_thr));
        };
        auto all_loud = [&]() {                                                // the largest quiet sum of this lane's SPL symbols
            uint32_t qm = 0;
#pragma unroll
            for (int piece = 0; piece < SPL; piece++) {
                uint32_t q = 0;
#pragma unroll
                for (int d = 0; d < NO; d++) q = quiet_sad(x[NO * piece + d], q);
                qm = q > qm ? q : qm;
            }
            return loud_enough(qm, (uint32_t)BF, amp_thr);
        };
        if constexpr (SPL == 1) {
            const int nv = (Kr - k0) < 64 ? (Kr - k0) : 64;
            rxd_pass<64>(rd, B[0], nv, k0, lane, words, out_row, out_stride, [&]() { return amp_word(0); });
        } else {
            rxd_round<SPL>(rd, B, Kr, k0, lane, words, out_row, out_stride, amp_word, all_loud);
        }
        if (rd.st.phase == 2) break;
        if (HINTED && partial) {           // no squelch stop among the symbols that were there: fetch the rest, run the round again
            rd = saved;
            fr.template fetch_through<(FLAGS & 4) ? 0 : 2>(last >> 10);
            r--;
        }
    }
}

}  // namespace afsk
