// afsk_demod_rt.h -- part of the single-pass demodulator (afsk_demod_fast.h includes the parts in order; see its header
// comment for the overall design): every other valid bit_frames as a RUN-TIME value: geometry, clock recovery, round loop, demod_stream_rt.
#pragma once

namespace afsk {

// ---- every other valid bit_frames (a RUNTIME value) on the single-pass ring -------------------
// Every multiple of 4 without a compile-time geometry: values no Receiver can have (bit_frames must
// divide 48000) but the C-ABI accepts, and -- inside a MIXED-baud launch -- the 17 general-piece rates.
// Same ring, watermark refill and mirror as wm_rounds / gp_rounds, geometry computed at run time:
//   * rounds: the general-piece scheme of gp_rounds with run-time values (since r3; the r2 form split a
//     symbol into 2^k EQUAL whole-dword pieces inside a quarter, which left bit_frames that are not a
//     multiple of 8 with two lanes per symbol -- 16 of 64 lanes busy): 4 ... 64 lanes per symbol at dword
//     granularity, NB dwords per lane against its constant template + two tail slots, all 64 lanes busy
//     for every bit_frames; the NB loop has a run-time trip count (four reads in flight per step);
//   * clock recovery: the sub-window form in steps of 64 x 24 offsets with run-time lags (seven
//     2-byte-aligned 48-byte sub-windows per lane and step), run twice -- once for the minimum, once
//     for the first offset under the bound -- because the totals of a run-time number of steps
//     cannot stay in registers.
typedef u32x4 u32x4_a2 __attribute__((aligned(2)));
typedef uint32_t u32_a2 __attribute__((aligned(2)));

// bit (g * lps + lps - 1) of a wave-uniform mask -> bit g (the last lane of every group)
__device__ __forceinline__ uint64_t compress_bits_last_rt(uint64_t x, int lps) {
    switch (lps) {
        case 4: return compress_bits_last<4>(x);
        case 8: return compress_bits_last<8>(x);
        case 16: return compress_bits_last<16>(x);
        case 32: return compress_bits_last<32>(x);
        default: return compress_bits_last<64>(x);
    }
}

// sum over the lps lanes of an aligned group, valid in the group's last lane
__device__ __forceinline__ uint32_t group_sum_last_rt(uint32_t v, int lps) {
    v = quad_sum<4>(v);
    if (lps >= 8) v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true);   // row_half_mirror
    if (lps >= 16) v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, true);  // row_mirror
    if (lps >= 32) v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false); // row_bcast:15 -> rows 1, 3
    if (lps >= 64) v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false); // row_bcast:31 -> rows 2, 3
    return v;
}

// rxd_pass with a run-time pass size ps (a power of two below 64; k0 % ps == 0)
template <class AmpFn>
__device__ __forceinline__ void rxd_pass_rt(RxDeferred& d, uint64_t bmask, int nv, int k0, int ps, int lane,
                                            unsigned long long* words, uint8_t* out_row, int out_stride,
                                            AmpFn&& amp_ok_mask) {
    const int start = rx_training(d.st, bmask, nv, k0);
    if (start >= 0 && start < nv) rxd_stop(d, amp_ok_mask(), start, nv, k0);
    const uint64_t valid = nv >= 64 ? ~0ull : ((1ull << nv) - 1ull);
    d.filled = k0 + ps;
    d.cur |= (bmask & valid) << (k0 & 63);
    if (((k0 & 63) + ps) == 64) {
        if (lane == 0) words[(k0 >> 6) & (kBitWords - 1)] = d.cur;
        d.cur = 0;
    }
    if (rxd_flush_due(d, k0 + nv)) rxd_flush<32>(d, k0 + nv, lane, words, out_row, out_stride);
}

// ONE sweep over the offsets: every lane keeps the first offset of its own minimal truncated mean --
// a later total replaces the candidate only if it lies below the lower edge of the candidate's bin
// (strictly smaller mean; equal means keep the earlier index, ref:332-337), so the division runs only
// on the rare updates -- and the wave minimum of (mean << 12 | index) is the reference's first index of
// the minimum.  (r2 swept twice, for the minimum and for the first offset under the bound.)  Sub-windows
// are read from the 4-byte-aligned address below their lag and shifted by 0 or 2 bytes in registers:
// 2-byte-aligned ds_read_b128 execute on gfx950, but several times slower.
__device__ __forceinline__ int recover_clock_index_rt(FastRing& fr, int bf) {
    constexpr int GC = 24, STEP = 64 * GC;
    const int lane = fr.lane;
    const int N = 2 * bf, q = bf >> 2, h = bf >> 1, NOFF = kSync - N;
    fr.template wait_exact<kRingChunks - 8>(7);                 // chunks 0..7 (samples 0..4095) have landed
    // total(0) = 65535 * bf + sum_j sigma_j x[j] over the 2*bf template samples (ref:80-91): dword m =
    // samples 2m, 2m + 1, lanes stride through the bf dwords
    uint32_t base;
    {
        const float rcp_q = 1.0f / (float)q;
        int32_t a = 0;
        for (int m = lane; m < bf; m += 64) {
            const uint32_t w = *reinterpret_cast<const uint32_t*>(fr.ring + 4 * m);
            uint32_t cf = 0;
#pragma unroll
            for (int half = 0; half < 2; half++) {
                const int j = 2 * m + half;
                const bool hi = j < bf ? ((div_exact((uint32_t)j, (uint32_t)q, rcp_q) & 1u) == 0) : ((j - bf) < h);
                cf |= (hi ? 0xFFFFu : 0x0001u) << (16 * half);    // sigma = -1 where the template is 32767
            }
            a = dot2_i16(w, cf, a);
        }
        const int32_t sum = __builtin_amdgcn_readlane(wave_incl_scan_dpp(a), 63);
        base = 65535u * (uint32_t)bf + (uint32_t)sum;
    }
    const int T = (NOFF + STEP - 1) / STEP;
    const int lag[7] = {0, q, 2 * q, 3 * q, bf, bf + h, N};
    constexpr int coef[7] = {1, -2, 2, -2, 2, -2, 1};
    const float rcp_n = 1.0f / (float)N;
    uint32_t lane_bound = 0xFFFFFFFFu, lane_key = 0xFFFFFFFFu;
    for (int t = 0; t < T; t++) {
        const int f = STEP * t + GC * lane;                     // even
        const int fa = f < NOFF ? f : NOFF - 2;                 // lanes past the last offset read inside the window (even too)
        const uint8_t* src = fr.ring + 2 * fa;                  // 4-byte aligned
        uint32_t R[7][GC / 2];
#pragma unroll
        for (int e = 0; e < 7; e++) {
            const uint8_t* p = src + 2 * (lag[e] & ~1);
            const uint32_t sh = (lag[e] & 1) ? 2u : 0u;           // wave-uniform
            uint32_t W[GC / 2 + 1];                               // dword reads (they pair up as ds_read2_b32): a 16-byte
#pragma unroll                                                    // read at a 4-byte-aligned address is a slow one
            for (int j = 0; j <= GC / 2; j++) W[j] = *reinterpret_cast<const uint32_t*>(p + 4 * j);
#pragma unroll
            for (int j = 0; j < GC / 2; j++) R[e][j] = __builtin_amdgcn_alignbyte(W[j + 1], W[j], sh);
        }
        int32_t run[GC];                                        // run[k] = total(f + k + 1) - total(f)
        int32_t acc = 0;
#pragma unroll
        for (int k = 0; k < GC; k++) {
#pragma unroll
            for (int e = 0; e < 7; e++) {
                const uint32_t c = (uint32_t)(uint16_t)(int16_t)coef[e];
                acc = dot2_i16(R[e][k >> 1], (k & 1) ? (c << 16) : c, acc);
            }
            run[k] = acc;
        }
        const int32_t incl = wave_incl_scan_dpp(acc);
        const uint32_t first = base + (uint32_t)(incl - acc);
        base += (uint32_t)__builtin_amdgcn_readlane(incl, 63);
#pragma unroll
        for (int k = 0; k < GC; k++) {
            const uint32_t tot = k == 0 ? first : first + (uint32_t)run[k - 1];
            const int i = f + k;
            if (i < NOFF && tot < lane_bound) {                  // strictly smaller mean than the lane's candidate
                const uint32_t mean = div_exact(tot, (uint32_t)N, rcp_n);
                lane_bound = mean * (uint32_t)N;
                lane_key = (mean << 12) | (uint32_t)i;
            }
        }
    }
    return (int)(wave_min_u32(lane_key) & 4095u);               // first index of the minimal mean (ref:332-337)
}

// geometry of the run-time general pieces (wave-uniform)
struct RtGeom {
    int lps, lpq_shift, spp, rbytes, nb;
};
__device__ __forceinline__ RtGeom rt_geometry(int bf) {
    RtGeom g;
    g.lps = 4;
    while (g.lps < 64 && (64 / g.lps) * 2 * bf > 7680) g.lps *= 2;
    g.lpq_shift = __builtin_ctz((unsigned)g.lps) - 2;             // lanes per quarter = 1 << lpq_shift
    g.spp = 64 / g.lps;
    g.rbytes = g.spp * 2 * bf;
    const int q = bf >> 2;
    int mn = 1 << 30;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int nk = (((k + 1) * q + 1) >> 1) - ((k * q + 1) >> 1);
        mn = (nk >> g.lpq_shift) < mn ? (nk >> g.lpq_shift) : mn;
    }
    g.nb = mn - 1;                                                // every lane: nb + 1 or nb + 2 dwords (>= 1: bit_frames >= 28)
    return g;
}

template <int FLAGS, bool HINTED>
__device__ __forceinline__ void rt_rounds(FastRing& fr, int bf, const RtGeom& g, int byte0, int32_t K,
                                          int32_t NR, uint32_t amp_thr, RxDeferred& rd,
                                          unsigned long long* words, uint8_t* out_row, int out_stride,
                                          int32_t* margins, int32_t mstride) {
    const int lane = fr.lane;
    const int q = bf >> 2;
    const int lps = g.lps, spp = g.spp, rbytes = g.rbytes, nb = g.nb;
    const int part = lane & (lps - 1), sym = lane / lps;
    const int32_t mlim = K < mstride ? K : mstride;
    const float rcp_bf = 1.0f / (float)bf;
    constexpr uint32_t FULL = 65535u;
    // this lane's piece of every symbol it works on: dwords [d0, d1) of quarter k (see gp_rounds)
    const int k = part >> g.lpq_shift, j = part & ((1 << g.lpq_shift) - 1);
    const int q0 = (k * q + 1) >> 1, q1 = ((k + 1) * q + 1) >> 1;
    const int d0 = q0 + ((j * (q1 - q0)) >> g.lpq_shift), d1 = q0 + (((j + 1) * (q1 - q0)) >> g.lpq_shift);
    const bool two = (d1 - d0) == nb + 2;
    const bool mark_hi = (k & 1) == 0, space_hi = k < 2;          // quarter k: mark hi,lo,hi,lo (ref:80-85) / space hi,hi,lo,lo (ref:68-77)
    const uint32_t cm = mark_hi ? 0xFFFFFFFFu : 0u, cs = space_hi ? 0xFFFFFFFFu : 0u;
    const int kl = (int)div_exact((uint32_t)(2 * d1 - 1), (uint32_t)q, 1.0f / (float)q);   // quarter of the piece's very last sample
    const uint32_t lm = (cm & 0xFFFFu) | (((kl & 1) == 0 ? 0xFFFFu : 0u) << 16);
    const uint32_t ls = (cs & 0xFFFFu) | ((kl < 2 ? 0xFFFFu : 0u) << 16);
    const uint32_t tmA = two ? cm : lm, tsA = two ? cs : ls;      // tail slot A = dword nb of the piece
    const int piece_byte = sym * 2 * bf + 4 * d0;
    int pos = byte0;                                              // stream byte of the round's first sample
    for (int r = 0; r < NR; r++, pos += rbytes) {
        const int last = pos + rbytes + 3;                        // tail slot B of the last lane reaches one dword further
        if constexpr (HINTED) fr.template fetch_through<(FLAGS & 4) ? 0 : 2>(last >> 10);
        fr.wait_landed(last >> 10);
        if constexpr (HINTED) fr.eval_probes(last >> 10, amp_thr / (uint32_t)bf, byte0, 4);
        const int rb = pos & (kRingBytes - 1);
        if (rb + rbytes + 4 > kRingBytes) {                       // a piece runs past the ring end: refresh the mirror
            if (lane < kMirrorBytes / 16)
                *reinterpret_cast<u32x4*>(fr.ring + kRingBytes + 16 * lane) =
                    *reinterpret_cast<const u32x4*>(fr.ring + 16 * lane);
            wave_lds_sync();
        }
        // linear from the dword-aligned address at or below the piece (an odd clock index puts it 2 bytes into
        // a dword; the mirror covers a piece): dword reads the compiler may pair up but never merges into reads
        // wider than their real alignment (gp_rounds has the story), shifted by 0 or 2 bytes in registers
        const uint32_t sh = (uint32_t)(byte0 & 2);                // wave-uniform
        const uint8_t* src = fr.ring + (((rb + piece_byte) & (kRingBytes - 1)) - (int)sh);
        uint32_t h = 0, amp = 0;
        int d = 0;
        for (; d + 4 <= nb; d += 4) {                             // five reads in flight per step
            uint32_t W[5];
#pragma unroll
            for (int u = 0; u < 5; u++) W[u] = *reinterpret_cast<const uint32_t*>(src + 4 * (d + u));
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t x = __builtin_amdgcn_alignbyte(W[u + 1], W[u], sh);
                h = __builtin_amdgcn_sad_u16(limit_pair_biased(x), 0xFFFFFFFFu, h);         // ref:344, 346-347
                amp = quiet_sad(x, amp);                                                    // ref:94-98 (quiet sum: see quiet_sad)
            }
        }
        uint32_t prev = *reinterpret_cast<const uint32_t*>(src + 4 * d);
        for (; d < nb; d++) {
            const uint32_t next = *reinterpret_cast<const uint32_t*>(src + 4 * d + 4);
            const uint32_t x = __builtin_amdgcn_alignbyte(next, prev, sh);
            prev = next;
            h = __builtin_amdgcn_sad_u16(limit_pair_biased(x), 0xFFFFFFFFu, h);
            amp = quiet_sad(x, amp);
        }
        const uint32_t wa = *reinterpret_cast<const uint32_t*>(src + 4 * nb + 4);
        const uint32_t wb = *reinterpret_cast<const uint32_t*>(src + 4 * nb + 8);
        const uint32_t xa = __builtin_amdgcn_alignbyte(wa, prev, sh);
        const uint32_t xb = __builtin_amdgcn_alignbyte(wb, wa, sh);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the round's reads have returned: refill
        fr.template top_up<(FLAGS & 4) ? 0 : 2, HINTED>(((pos + rbytes) >> 10) + kRingChunks);
        const uint32_t la = limit_pair_biased(xa), lb = limit_pair_biased(xb);
        uint32_t mark = mark_hi ? h : FULL * (2u * (uint32_t)nb) - h;
        uint32_t space = space_hi ? h : FULL * (2u * (uint32_t)nb) - h;
        mark = __builtin_amdgcn_sad_u16(la, tmA, mark);
        space = __builtin_amdgcn_sad_u16(la, tsA, space);
        amp = quiet_sad(xa, amp);
        const uint32_t mb = __builtin_amdgcn_sad_u16(lb, lm, 0u), sb = __builtin_amdgcn_sad_u16(lb, ls, 0u);
        const uint32_t ab = quiet_sad(xb, 0u);
        mark += two ? mb : 0u;
        space += two ? sb : 0u;
        amp += two ? ab : 0u;
        mark = group_sum_last_rt(mark, lps);
        space = group_sum_last_rt(space, lps);
        amp = group_sum_last_rt(amp, lps);
        const int k0 = r * spp;
        const uint32_t md = div_exact(mark, (uint32_t)bf, rcp_bf), sd = div_exact(space, (uint32_t)bf, rcp_bf);
        const bool bit = md < sd;                                            // ref:348-351 (read from the group's last lane)
        if (margins && part == lps - 1 && k0 + sym < mlim) margins[k0 + sym] = (int32_t)sd - (int32_t)md;
        const int nv = (K - k0) < spp ? (K - k0) : spp;
        const uint64_t bmask = compress_bits_last_rt(__ballot(bit), lps);
        rxd_pass_rt(rd, bmask, nv, k0, spp, lane, words, out_row, out_stride, [&]() {
            return compress_bits_last_rt(__ballot(loud_enough(amp, (uint32_t)bf, amp_thr)), lps);
        });
        if (rd.st.phase == 2) break;
    }
}

template <int FLAGS, bool BIG = true>
__device__ __forceinline__ void demod_stream_rt(const int16_t* xs, int32_t len, int bf, int32_t amp_end,
                                                uint8_t* lds, int lane, RxState& st, uint8_t* out_row,
                                                int out_stride, int& ci_out, int32_t& n_sym_out,
                                                int32_t* margins, int32_t mstride, bool warm_arg, bool hint_arg) {
    const bool warm = BIG && warm_arg, hint = BIG && hint_arg;
    FastRing fr;
    fr.rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)xs, 0, len * 2, 0x00020000);
    fr.ring = lds;
    fr.lane = lane;
#pragma unroll
    for (int c = 0; c < kRingChunks; c++) fr.template issue<(FLAGS & 4) ? 0 : 2>(c);
    fr.next = kRingChunks;
    if (warm) {
#pragma unroll
        for (int p = 0; p < kWarmOps; p++)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(fr.rsrc, AFSK_LDS(lds + kWarmDummyOffset), 4, lane * 64,
                                                     kRingBytes + 4096 * p, 0, 0);
        fr.warm_ops = kWarmOps;
    }
    int ci = 0;
    if constexpr (FLAGS & 1) fr.template wait_exact<kRingChunks - 8>(7);
    else ci = recover_clock_index_rt(fr, bf);
    ci_out = ci;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const RtGeom g = rt_geometry(bf);                           // lanes per symbol, symbols per round, dwords per lane
    const int spp = g.spp;
    const int32_t K = (len - ci - 1) / bf;                      // symbols with i < len - bf (ref:362,372)
    n_sym_out = K;
    const int32_t NR = (K + spp - 1) / spp;
    const uint32_t amp_thr =
        (uint32_t)(amp_end < 0 ? 0 : (amp_end > 40000 ? 40000 : amp_end)) * (uint32_t)bf;
    const int byte0 = 2 * ci;
    if (hint) fr.request_probes((uint32_t)len * 2u, byte0, g.rbytes);   // tail hint (see kProbes)
    {
        const int lim = (byte0 >> 10) + kRingChunks;            // chunks entirely below the clock index are free
        while (fr.next < lim) { fr.template issue<(FLAGS & 4) ? 0 : 2>(fr.next); fr.next++; }
    }
    unsigned long long* words = reinterpret_cast<unsigned long long*>(lds + kBitBufOffset);
    RxDeferred rd;
    rxd_init(rd);
    if (BIG && hint) rt_rounds<FLAGS, BIG>(fr, bf, g, byte0, K, NR, amp_thr, rd, words, out_row, out_stride, margins, mstride);
    else rt_rounds<FLAGS, false>(fr, bf, g, byte0, K, NR, amp_thr, rd, words, out_row, out_stride, margins, mstride);
    rxd_finish<32>(rd, K, lane, words, out_row, out_stride);
    st = rd.st;
    wait_vmcnt<0>();   // drain DMA still in flight before the LDS region is released
}

}  // namespace afsk
