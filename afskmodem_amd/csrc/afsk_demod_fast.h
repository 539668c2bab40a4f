// afsk_demod_fast.h -- stream-aligned single-pass path.  Included by afsk_demod_impl.h.
// Covers every bit_frames value of the reference's documented 300 - 12000 baud range:
//   20 / 40 / 80 / 160   (2400 / 1200 / 600 / 300 baud)        fast_rounds: 5 KiB rounds, 80-byte lane pieces
//   240 / 320 / 480      (200 / 150 / 100 baud: below the documented range, but on the list of rates the
//                         reference's code round-trips)         wm_rounds with 4 / 8 lanes per symbol
//   4 / 8 / 12 / 16 / 24 / 32 / 48 / 64  (12000 ... 750 baud)  multi_rounds: rounds of whole chunks (5 / 5 / 6 / 8 / 6 / 4 / 6 / 8), several
//                                                              symbols per lane
//   60 / 96 / 100 / 120  (800 / 500 / 480 / 400 baud)          wm_rounds: rounds of any size, watermark refill
// and every other valid bit_frames -- every multiple of 4 below 2048 that is not listed above: 28, 36,
// 44 ... 124 (reachable as int(48000 / baud)) and 128 and above (375 baud and below, outside the
// documented range but decodable by the reference's code) -- with a geometry computed at run time
// (rt_rounds).
//
// Every sample is fetched from HBM exactly once: the wave starts a 16 KiB LDS-DMA
// ring at sample 0 the moment it starts, BEFORE the clock index is known, so the
// memory pipe is busy while phase A computes.
//
//   ring      16 x 1 KiB chunks; chunk c (samples 512c .. 512c+511) lives in slot
//             c & 15; one `buffer_load_dwordx4 ... lds` per chunk, bounds-checked by
//             a descriptor over the whole stream (tail reads return 0).  All 16 chunks
//             are requested at once, before anything else.
//   phase A   ref:322-339 once chunks 0..7 have landed: lane-wise sliding correlation
//             (recover_clock_index_lanes for bit_frames <= 120, _lane_steps for 160): every lane keeps the raw samples of
//             its own run of consecutive sync offsets in registers and slides the SAD with 7
//             v_dot2c_i32_i16 per offset.  (The first design -- a DPP prefix-sum producer feeding a
//             circular LDS window, 7 lookups per offset -- was LDS-bandwidth bound and is gone.)
//   re-base   (r6) once the clock index is known the ring contents and the buffer descriptor move by the whole dwords
//             of (2*ci) & 15, so that every stream runs one of two forms of its round loop, both with the tail hint
//   phase B   ref:342-351: every lane owns an 80-byte piece (40 samples: one
//             1200-baud symbol, two 2400-baud symbols, half a 600-baud or a quarter of a
//             300-baud symbol) at ring byte (byte0 + 5120*r + 80*lane) mod 16 KiB, byte0 = the ring byte of
//             symbol 0 after the stream has been re-based on the clock index (FastRing::rebase, r6: a multiple
//             of 16, or 2 bytes past one for an odd clock index): five aligned ds_read_b128, or -- the ODD form --
//             six, shifted down by one sample in registers (v_alignbyte).  After the reads of round r the five
//             chunks it consumed are refilled immediately (11 KiB stay in flight).
//   phase C   ref:361-378, 145-163, 393-399: terminator scan and squelch stop on wave-uniform
//             ballot masks; the Hamming decode + byte pack is deferred and vectorised.
//
// The code is split over afsk_demod_ring.h (the ring), afsk_demod_sync.h (phase A), afsk_demod_phasec.h (phase C),
// afsk_demod_rounds_{fast,multi,wm,gp}.h (one round loop per geometry family), afsk_demod_rt.h (run-time geometry);
// this file holds demod_stream_fast: one stream, start to finish, by one wave.
#pragma once
#include <type_traits>

#include "afsk_demod_ring.h"
#include "afsk_demod_sync.h"
#include "afsk_demod_phasec.h"
#include "afsk_demod_rounds_fast.h"
#include "afsk_demod_rounds_multi.h"
#include "afsk_demod_rounds_wm.h"
#include "afsk_demod_rounds_gp.h"
#include "afsk_demod_rt.h"

namespace afsk {

template <int BF, int FLAGS, bool BIG = true, bool UNIFORM = false>
__device__ __forceinline__ void demod_stream_fast(const int16_t* xs, int32_t len, int32_t amp_end,
                                                  uint8_t* lds, int lane, RxState& st,
                                                  uint8_t* out_row, int out_stride, int& ci_out,
                                                  int32_t& n_sym_out,
                                                  unsigned long long* stamps = nullptr,
                                                  int32_t* margins = nullptr, int32_t mstride = 0,
                                                  bool warm_arg = false, bool hint_arg = false) {
    // BIG = false: the copy of this code that small launches run -- no warming, no hint, not even
    // their tests (the extra live scalars cost config #2 1.5 % when both lived in one copy)
    const bool warm = BIG && warm_arg, hint = BIG && hint_arg;
    constexpr bool MULTI = MultiGeom<BF>::valid;
    constexpr bool WM = WmGeom<BF>::valid;
    constexpr bool GP = !MULTI && !WM && BF != 20 && BF != 40 && BF != 80 && BF != 160;   // general pieces
    constexpr int SPR = MULTI ? MultiGeom<BF>::SPR : (WM ? WmGeom<BF>::SPP : (GP ? GpGeom<GP ? BF : 128>::SPP : 2560 / BF));   // symbols per round
    FastRing fr;
    fr.rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)xs, 0, len * 2, 0x00020000);
    fr.ring = lds;
    fr.lane = lane;
    // the lane-wise clock recovery needs no LDS of its own: the whole ring is requested at once
    constexpr int PRE = kRingChunks;
    fr.template issue_run<(FLAGS & 4) ? 0 : 2, PRE>(0);
    fr.next = PRE;
    if (warm) {
#pragma unroll
        for (int p = 0; p < kWarmOps; p++)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(fr.rsrc, AFSK_LDS(lds + kWarmDummyOffset), 4, lane * 64,
                                                     kRingBytes + 4096 * p, 0, 0);
        fr.warm_ops = kWarmOps;
    }

    int ci = 0;
    if constexpr (FLAGS & 1) {
        fr.template wait_exact<PRE - 8>(7);
    } else {
        // contiguous lane windows wherever the register file takes them (a 300-baud lane window does
        // not: 72 + 320 samples), sub-windows in steps otherwise
        constexpr bool LANES_FORM = BF <= 120;         // 160 / 240 / 320 / 480: sub-windows in steps
#ifndef AFSK_NO_ZERO_SHORTCUT
        if (clock_index_is_zero<BF, PRE>(fr))          // the training sequence starts at sample 0: no search (ref:332-337)
            ci = 0;
        else
#endif
        if constexpr (LANES_FORM)
            ci = recover_clock_index_lanes<BF, false, PRE>(fr, nullptr, (FLAGS & 64) ? stamps : nullptr);
        else
            ci = recover_clock_index_lane_steps<BF, false, PRE>(fr, nullptr, (FLAGS & 64) ? stamps : nullptr);
    }
    ci_out = ci;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if constexpr (FLAGS & 64) {
        if (lane == 0) stamps[1] = __builtin_amdgcn_s_memrealtime();
    }

    const int32_t rel_len = len - ci;
    const int32_t K = (rel_len - 1) / BF;                      // symbols with i < len - bf (ref:362,372)
    n_sym_out = K;
    const int32_t NR = (K + SPR - 1) / SPR;
    const uint32_t amp_thr =
        (uint32_t)(amp_end < 0 ? 0 : (amp_end > 40000 ? 40000 : amp_end)) * (uint32_t)BF;
    // r6: the stream is re-based on the clock index (FastRing::rebase): symbol 0 sits at a 16-byte-aligned ring byte, or
    // 2 bytes past one when ci is odd -- two forms of every round loop instead of a re-aligning copy with seven shifts
    // behind a switch, and the tail hint for every stream (until r5: none when (2 * ci) & 15 != 0)
    const int byte0 = fr.template rebase<(FLAGS & 4) ? 0 : 2>(xs, len, 2 * ci);   // ring byte of symbol 0
    const bool odd = (byte0 & 2) != 0;
    // Tail hint (see kProbes): the large-launch kernels (BIG), for every stream.  The UNIFORM kernels of
    // bit_frames 4 / 8 take it from kHintMinStreamsShort4 / 8 streams on (at 4096 streams it costs them 3 %).
    // (Until r5 those two went without it inside the per-stream kernel -- five- and ten-slice rounds, short of scalar
    // registers: with the r5 build the spills are gone and the hint is worth 8 % there: profiles/r5_exp18_*.)
    const bool hinted = hint;
    if (hinted) {
        constexpr int kRoundBytes = MULTI ? 1024 * MultiGeom<MULTI ? BF : 4>::R : (WM ? WmGeom<WM ? BF : 60>::RBYTES : (GP ? GpGeom<GP ? BF : 128>::RBYTES : 5120));
        fr.request_probes((uint32_t)len * 2u - (uint32_t)(2 * ci - byte0), byte0, kRoundBytes);
    }
    // chunks entirely below the clock index are free already
    {
        const int lim = (byte0 >> 10) + kRingChunks;
        while (fr.next < lim) { fr.template issue<(FLAGS & 4) ? 0 : 2>(fr.next); fr.next++; }
    }
    // phase C state; its bit buffer reuses the LDS behind the ring (phase A's window is done)
    constexpr int PS = SPR < 64 ? SPR : 64;                      // symbols per rxd pass
    unsigned long long* words = reinterpret_cast<unsigned long long*>(lds + kBitBufOffset);
    RxDeferred rd;
    rxd_init(rd);
    // (a BIG kernel is only ever launched with the hint on -- launch_demod / launch_demod_uniform pick it by the same
    // stream count -- so it holds the hinted forms alone, a small-launch kernel the plain ones)
#define AFSK_ROUNDS(FN)                                                                                              \
    do {                                                                                                             \
        if (odd) FN<BF, FLAGS, true, BIG>(fr, byte0, K, NR, amp_thr, rd, words, out_row, out_stride, margins, mstride); \
        else FN<BF, FLAGS, false, BIG>(fr, byte0, K, NR, amp_thr, rd, words, out_row, out_stride, margins, mstride);  \
    } while (0)
    if constexpr (GP) AFSK_ROUNDS(gp_rounds);
    else if constexpr (WM) AFSK_ROUNDS(wm_rounds);
    else if constexpr (MULTI) AFSK_ROUNDS(multi_rounds);
    else AFSK_ROUNDS(fast_rounds);
#undef AFSK_ROUNDS
    rxd_finish<PS>(rd, K, lane, words, out_row, out_stride);
    st = rd.st;
    wait_vmcnt<0>();   // drain DMA still in flight before the LDS region is released
}

}  // namespace afsk
