// afsk_demod_rounds_fast.h -- part of the single-pass demodulator (afsk_demod_fast.h includes the parts in order; see its header
// comment for the overall design): round loop of bit_frames 20 / 40 / 80 / 160 (2400 / 1200 / 600 / 300 baud): 5 KiB rounds, 80-byte lane pieces.
#pragma once

namespace afsk {

// One 5 KiB round: symbol decisions, then (only once the training terminator has been
// seen) squelch amplitudes, Hamming decode and byte pack.
template <int BF, int FLAGS>
__device__ __forceinline__ void fast_round_compute(const uint32_t (&x)[20], int lane, uint32_t amp_thr,
                                                   int32_t K, int k0, RxDeferred& rd,
                                                   unsigned long long* words, uint8_t* out_row,
                                                   int out_stride, int32_t* margins, int32_t mstride) {
    constexpr int Q = BF / 4, H = BF / 2;
    const int32_t mlim = K < mstride ? K : mstride;      // soft output rows hold symbols [0, mlim)
    constexpr uint32_t FULL = 65535u;
    if constexpr (BF == 40) {                 // one symbol per lane, 5 dwords per quarter
        uint32_t mark, space;
        if constexpr (FLAGS & 2) {
            uint32_t o = 0;
#pragma unroll
            for (int d = 0; d < 20; d++) o |= x[d];
            mark = o & 1; space = 1;
        } else {
            const uint32_t h0 = hi_sad<0, 5>(x), h1 = hi_sad<5, 10>(x), h2 = hi_sad<10, 15>(x),
                           h3 = hi_sad<15, 20>(x);
            // mark = hi,lo,hi,lo quarters (ref:80-85); space = hi,hi,lo,lo (ref:68-77)
            mark = 2u * FULL * Q + h0 + h2 - h1 - h3;
            space = 2u * FULL * Q + h0 + h1 - h2 - h3;
        }
        const uint32_t md = mark / (uint32_t)BF, sd = space / (uint32_t)BF;
        const bool bit = md < sd;                                            // ref:348-351
        if (margins && k0 + lane < mlim) margins[k0 + lane] = (int32_t)sd - (int32_t)md;
        const int nv = (K - k0) < 64 ? (K - k0) : 64;
        rxd_pass<64>(rd, __ballot(bit), nv, k0, lane, words, out_row, out_stride, [&]() {
            uint32_t q = 0u;                                    // (FLAGS & 2, a kbench ablation: always loud)
            if constexpr (!(FLAGS & 2)) q = quiet_sum<0, 20>(x);
            return __ballot(loud_enough(q, (uint32_t)BF, amp_thr));
        });
    } else if constexpr (BF == 20) {          // two symbols per lane: dwords 0-9 = symbol k0 + lane,
                                              // dwords 10-19 = symbol k0 + 64 + lane (fast_rounds reads
                                              // the two 40-byte pieces), so each half is one plain ballot
        uint32_t mk[2] = {0, 0}, sp[2] = {0, 0};
#pragma unroll
        for (int d = 0; d < 20; d++) {
            const int h2 = d / 10, dd = d % 10;
            const uint32_t lim = limit_pair_biased(x[d]);
            const uint32_t tm = mark_half(2 * dd, Q) | (mark_half(2 * dd + 1, Q) << 16);
            const uint32_t ts = space_half(2 * dd, H) | (space_half(2 * dd + 1, H) << 16);
            mk[h2] = __builtin_amdgcn_sad_u16(lim, tm, mk[h2]);
            sp[h2] = __builtin_amdgcn_sad_u16(lim, ts, sp[h2]);
        }
        const uint32_t md0 = mk[0] / (uint32_t)BF, sd0 = sp[0] / (uint32_t)BF;
        const uint32_t md1 = mk[1] / (uint32_t)BF, sd1 = sp[1] / (uint32_t)BF;
        const bool bit[2] = {md0 < sd0, md1 < sd1};
        if (margins) {
            if (k0 + lane < mlim) margins[k0 + lane] = (int32_t)sd0 - (int32_t)md0;
            if (k0 + 64 + lane < mlim) margins[k0 + 64 + lane] = (int32_t)sd1 - (int32_t)md1;
        }
        const uint64_t B[2] = {__ballot(bit[0]), __ballot(bit[1])};
        rxd_round<2>(rd, B, K, k0, lane, words, out_row, out_stride, [&](int half) {
            const uint32_t q = half == 0 ? quiet_sum<0, 10>(x) : quiet_sum<10, 20>(x);
            return __ballot(loud_enough(q, (uint32_t)BF, amp_thr));
        }, [&]() {
            const uint32_t q0 = quiet_sum<0, 10>(x), q1 = quiet_sum<10, 20>(x);
            return loud_enough(q0 > q1 ? q0 : q1, (uint32_t)BF, amp_thr);
        });
    } else {                                  // BF = 80 / 160: two / four lanes per symbol
        static_assert(BF == 80 || BF == 160, "fast path supports bit_frames 20, 40, 80, 160");
        constexpr int LPS = BF / 40;                               // lanes per symbol
        constexpr int QPL = 4 / LPS;                               // quarters per lane (2 or 1)
        constexpr int DPQ = 20 / QPL;                              // dwords per quarter
        constexpr int SPP = 64 / LPS;                              // symbols per pass
        const int part = lane & (LPS - 1);
        // quarter qi of the symbol: mark template hi,lo,hi,lo (ref:80-85), space hi,hi,lo,lo (ref:68-77);
        // the SAD against a lo template is 65535 * Q minus the SAD against the hi template
        uint32_t mark, space;
        if constexpr (QPL == 1) {
            const uint32_t h = hi_sad<0, 20>(x), l = FULL * Q - h;
            mark = (part & 1) ? l : h;
            space = part < 2 ? h : l;
        } else {
            const uint32_t ha = hi_sad<0, DPQ>(x), hb = hi_sad<DPQ, 20>(x);
            mark = ha + (FULL * Q - hb);                           // quarters 2*part (even), 2*part+1 (odd)
            space = part == 0 ? ha + hb : 2u * FULL * Q - ha - hb;
        }
        // sum over the LPS lanes of a symbol with DPP quad permutes (VALU only; __shfl_xor would
        // be a ds_bpermute round trip through the LDS pipe per step)
        mark = quad_sum<LPS>(mark);
        space = quad_sum<LPS>(space);
        const uint32_t md = mark / (uint32_t)BF, sd = space / (uint32_t)BF;
        const bool bit = md < sd;                                  // same in all LPS lanes of the symbol
        if (margins && part == 0 && k0 + lane / LPS < mlim)
            margins[k0 + lane / LPS] = (int32_t)sd - (int32_t)md;
        const int nv = (K - k0) < SPP ? (K - k0) : SPP;
        // every LPS-th bit of the ballot, compacted on the scalar unit: bit j <- symbol j
        const uint64_t bmask = compress_bits<LPS>(__ballot(bit));
        rxd_pass<SPP>(rd, bmask, nv, k0, lane, words, out_row, out_stride, [&]() {
            const uint32_t q = quad_sum<LPS>(quiet_sum<0, 20>(x));
            return amp_ok_word<LPS, false>(__ballot(loud_enough(q, (uint32_t)BF, amp_thr)), [](uint64_t b) { return compress_bits<LPS>(b); });
        });
    }
}

// The round loop.  byte0 -- the ring byte of symbol 0 after FastRing::rebase (r6) -- is a multiple of 16 (ODD = false:
// five aligned ds_read_b128 feed the arithmetic directly) or, for an odd clock index, 2 bytes past one (ODD = true: six
// aligned reads, shifted down by the one sample with a v_alignbyte per dword -- neither the LDS-DMA nor an LDS read moves
// data by half a dword at speed).  (Until r5: seven shifts behind a switch in every round, and no tail hint for them.)
template <int BF, int FLAGS, bool ODD, bool HINTED>
__device__ __forceinline__ void fast_rounds(FastRing& fr, int byte0, int32_t K, int32_t NR,
                                            uint32_t amp_thr, RxDeferred& rd,
                                            unsigned long long* words, uint8_t* out_row,
                                            int out_stride, int32_t* margins, int32_t mstride) {
    constexpr int SPR = 2560 / BF;                                // symbols per 5 KiB round
    constexpr bool ALIGNED = !ODD;
    const int lane = fr.lane;
    for (int r = 0; r < NR; r++) {
        // bytes [byte0 + 5120 r, byte0 + 5120 (r+1)) must have landed: at most 6 chunks
        // (B_r .. B_r+5) from the oldest resident one; chunks through B_r+15 are issued, so the
        // 10 youngest DMAs may still be in flight.
        int32_t Kr = K;                    // symbols this round may use (fewer: a partial round, see holding_wait)
        bool partial = false;
        RxDeferred saved;
        const int last = byte0 + 5120 * r + 5119 + (ALIGNED ? 0 : 16);                 // last byte read
        if (HINTED && fr.hint_holding()) { // the tail hint has stopped the fixed 5-chunks-per-round schedule
            Kr = fr.template holding_wait<(FLAGS & 4) ? 0 : 2>(last, K, r * SPR, byte0, 2 * BF, partial);
            if (partial) saved = rd;
        } else {
            fr.template wait_fixed<10>(((byte0 + 5120 * r) >> 10) + 5);
            if constexpr (HINTED) fr.template eval_probes<fine_probes(5120)>(((byte0 + 5120 * r) >> 10) + 5, amp_thr / (uint32_t)BF, byte0, ALIGNED ? 0 : 16, 2 * BF);
        }
        uint32_t x[20];
        const int rb = (byte0 + 5120 * r) & (kRingBytes - 1);         // wave-uniform
        if constexpr (BF == 20) {
            // 2400 baud: lane l takes symbol l (bytes 40l .. 40l+39 of the round) and symbol 64 + l
            // (2560 bytes further): ten 8-byte reads; the 40-byte lane stride spreads 32 lanes over
            // all 64 banks.
#pragma unroll
            for (int piece = 0; piece < 2; piece++) {
                const int pb = rb + 2560 * piece + 40 * lane;
                if constexpr (ALIGNED) {
#pragma unroll
                    for (int j = 0; j < 5; j++) {
                        const u32x2 t2 = *reinterpret_cast<const u32x2*>(fr.ring + ((pb + 8 * j) & (kRingBytes - 1)));
                        x[10 * piece + 2 * j] = t2[0]; x[10 * piece + 2 * j + 1] = t2[1];
                    }
                } else {
                    const int ab = pb & ~7;
                    uint32_t W[12];
#pragma unroll
                    for (int j = 0; j < 6; j++) {
                        const u32x2 t2 = *reinterpret_cast<const u32x2*>(fr.ring + ((ab + 8 * j) & (kRingBytes - 1)));
                        W[2 * j] = t2[0]; W[2 * j + 1] = t2[1];
                    }
                    uint32_t y[10];
                    realign_n<2, 12, 10>(W, y);
#pragma unroll
                    for (int d = 0; d < 10; d++) x[10 * piece + d] = y[d];
                }
            }
        } else if constexpr (ALIGNED) {
            if (rb + 5120 <= kRingBytes) {                              // no wrap in this round
                const uint8_t* src = fr.ring + rb + 80 * lane;
#pragma unroll
                for (int j = 0; j < 5; j++) {
                    const u32x4 t4 = *reinterpret_cast<const u32x4*>(src + 16 * j);
                    x[4 * j] = t4[0]; x[4 * j + 1] = t4[1]; x[4 * j + 2] = t4[2]; x[4 * j + 3] = t4[3];
                }
            } else {
                const int pb = rb + 80 * lane;
#pragma unroll
                for (int j = 0; j < 5; j++) {
                    const u32x4 t4 = *reinterpret_cast<const u32x4*>(fr.ring + ((pb + 16 * j) & (kRingBytes - 1)));
                    x[4 * j] = t4[0]; x[4 * j + 1] = t4[1]; x[4 * j + 2] = t4[2]; x[4 * j + 3] = t4[3];
                }
            }
        } else {
            const int ab = (rb + 80 * lane) & ~15;
            uint32_t W[24];
#pragma unroll
            for (int j = 0; j < 6; j++) {
                const u32x4 t4 = *reinterpret_cast<const u32x4*>(fr.ring + ((ab + 16 * j) & (kRingBytes - 1)));
                W[4 * j] = t4[0]; W[4 * j + 1] = t4[1]; W[4 * j + 2] = t4[2]; W[4 * j + 3] = t4[3];
            }
            realign<2>(W, x);
        }
        // the reads above have returned (their values are in x): refill the 5 chunks this
        // round consumed right away, before the arithmetic
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (HINTED && partial) {
            // (no refill: the round may have to run again on the same ring contents)
        } else if (HINTED && fr.hint_takes_over(5)) {
            fr.template top_up<(FLAGS & 4) ? 0 : 2>(((byte0 + 5120 * (r + 1)) >> 10) + kRingChunks);
        } else {
            fr.template issue_run<(FLAGS & 4) ? 0 : 2, 5>(fr.next);
            fr.next += 5;
        }
        fast_round_compute<BF, FLAGS>(x, lane, amp_thr, Kr, r * SPR, rd, words, out_row, out_stride,
                                      margins, mstride);
        if (rd.st.phase == 2) break;
        if (HINTED && partial) {           // no squelch stop among the symbols that were there: fetch the rest, run the round again
            rd = saved;
            fr.template fetch_through<(FLAGS & 4) ? 0 : 2>(last >> 10);
            r--;
        }
    }
}

}  // namespace afsk
