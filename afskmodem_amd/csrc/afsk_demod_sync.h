// afsk_demod_sync.h -- part of the single-pass demodulator (afsk_demod_fast.h includes the parts in order; see its header
// comment for the overall design): phase A, clock recovery ref:322-339: the lane-wise sliding correlation (bit_frames <= 120) and its form
// in steps (160 and above).
#pragma once

namespace afsk {

// ---- phase A, lane-wise form (every single-pass bit_frames up to 120) --------------------
// Every lane owns GC = 72 CONSECUTIVE sync offsets and the GC + 2*BF raw samples they touch,
// loaded once from the ring into registers (14 / 19 / 29 ... 39 ds_read_b128 at bit_frames
// 20 / 40 / 80 ... 120; the 144-byte lane stride makes them bank-conflict free).  Against the full-scale square template no abs is needed:
//   total(i) = 65535*BF + sum_j sigma_j * x[i+j],  sigma_j = -1 where the template is 32767,
//                                                            +1 where it is -32768,
// so the first offset of a lane is N/2 v_dot2 and every further offset slides by
//   total(i+1) - total(i) = x[i] - 2x[i+Q] + 2x[i+2Q] - 2x[i+3Q] + 2x[i+BF] - 2x[i+BF+H] + x[i+N]
// = 7 v_dot2c_i32_i16 with a (coef, 0) / (0, coef) constant picking the half of the dword.
// No prefix sums, no cross-lane scan, no window in LDS: 28 KB of LDS reads per stream instead
// of ~130 KB, and ~40 % fewer VALU instructions than the prefix-window form above.
typedef short s16x2 __attribute__((ext_vector_type(2)));

template <int BF>
struct LaneSync {
    static constexpr int N = 2 * BF, Q = BF / 4, H = BF / 2;
    static constexpr int NOFF = kSync - N;                     // ref:327
    static constexpr int GC = 72;                              // offsets per lane
    static constexpr int WD = (GC + N) / 2;                    // dwords in a lane's sample window
    static constexpr int LANES = (NOFF + GC - 1) / GC;         // lanes that own valid offsets
    static constexpr int KMIN = NOFF - GC * (LANES - 1);       // last lane: offsets k >= KMIN are invalid
    static_assert(WD % 4 == 0 && (GC * 2) % 16 == 0, "window must be whole 16-byte reads");
    static_assert(LANES <= 64 && GC * (LANES - 1) * 2 + WD * 4 <= kRingBytes, "window outside the ring");
    // sigma of template position j (ref:80-91: mark = hi,lo,hi,lo quarters, then space = hi,lo halves)
    static constexpr int sigma(int j) {
        return j < BF ? ((((j / Q) & 1) == 0) ? -1 : 1) : ((j - BF) < H ? -1 : 1);
    }
    static constexpr uint32_t sigma_pair(int d) {
        return ((uint32_t)(uint16_t)(int16_t)sigma(2 * d)) | ((uint32_t)(uint16_t)(int16_t)sigma(2 * d + 1) << 16);
    }
};

__device__ __forceinline__ int32_t dot2_i16(uint32_t pair, uint32_t coef, int32_t acc) {
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2, pair), __builtin_bit_cast(s16x2, coef), acc, false);
}

template <int BF, bool DEBUG = false, int PRE = kRingChunks>
__device__ __forceinline__ int recover_clock_index_lanes(FastRing& fr, uint32_t* dbg = nullptr,
                                                         unsigned long long* stamps = nullptr) {
    using L = LaneSync<BF>;
    constexpr int N = L::N, Q = L::Q, H = L::H, GC = L::GC, WD = L::WD, NOFF = L::NOFF;
    constexpr uint32_t C = 65535u * (uint32_t)BF;
    // floor(m / N) = mul_hi(m, ceil(2^(32+SH) / N)) >> SH, exact while m * N < 2^(32+SH); m <= 65535 * N
    constexpr int SH = N <= 16 ? 0 : 4;
    constexpr uint32_t M = (uint32_t)(((1ull << (32 + SH)) + N - 1) / N);
    static_assert(N <= 512 && ((1ull << (32 + SH)) + N - 1) / N < (1ull << 32) &&
                  65535ull * N * N < (1ull << (32 + SH)), "magic divisor out of range");
    const int lane = fr.lane;
    using std::integral_constant;

    fr.template wait_exact<PRE - 8>(7);                       // chunks 0..7 (samples 0..4095) have landed
    if (stamps && lane == 0) stamps[2] = __builtin_amdgcn_s_memrealtime();
    const int ll = lane < L::LANES ? lane : L::LANES - 1;    // idle lanes re-read the last window
    const uint8_t* src = fr.ring + (GC * 2) * ll;
    uint32_t R[WD];
#pragma unroll
    for (int j = 0; j < WD / 4; j++) {
        const u32x4 t4 = *reinterpret_cast<const u32x4*>(src + 16 * j);
        R[4 * j] = t4[0]; R[4 * j + 1] = t4[1]; R[4 * j + 2] = t4[2]; R[4 * j + 3] = t4[3];
    }
    // first offset of the lane: the full 2*BF-sample correlation
    int32_t acc = 0;
    static_for<0, N / 2>([&](auto dc) {
        constexpr int d = decltype(dc)::value;
        acc = dot2_i16(R[d], L::sigma_pair(d), acc);
    });
    // lanes without valid offsets start far above any real total (|sum of deltas| < 2^25)
    uint32_t total = lane < L::LANES ? C + (uint32_t)acc : 0xF0000000u;
    const bool last_lane = lane >= L::LANES - 1;
    uint32_t totals[GC];
    uint32_t min_total = 0xFFFFFFFFu;
    static_for<0, GC>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        if constexpr (k > 0) {
            // x[m] = half (m & 1) of R[m >> 1]; coefficient placed in the matching half
            constexpr int i = k - 1;
            auto term = [&](auto mc, auto cc, int32_t a) {
                constexpr int m = decltype(mc)::value;
                constexpr int c = decltype(cc)::value;
                constexpr uint32_t coef = (m & 1) ? ((uint32_t)(uint16_t)(int16_t)c << 16) : (uint32_t)(uint16_t)(int16_t)c;
                return dot2_i16(R[m >> 1], coef, a);
            };
            int32_t dl = 0;
            dl = term(integral_constant<int, i>{}, integral_constant<int, 1>{}, dl);
            dl = term(integral_constant<int, i + Q>{}, integral_constant<int, -2>{}, dl);
            dl = term(integral_constant<int, i + 2 * Q>{}, integral_constant<int, 2>{}, dl);
            dl = term(integral_constant<int, i + 3 * Q>{}, integral_constant<int, -2>{}, dl);
            dl = term(integral_constant<int, i + BF>{}, integral_constant<int, 2>{}, dl);
            dl = term(integral_constant<int, i + BF + H>{}, integral_constant<int, -2>{}, dl);
            dl = term(integral_constant<int, i + N>{}, integral_constant<int, 1>{}, dl);
            total += (uint32_t)dl;
        }
        uint32_t t = total;
        if constexpr (k >= L::KMIN) t = last_lane ? 0xFFFFFFFFu : t;      // offsets >= 4096 - 2*BF
        if constexpr (DEBUG) { if (GC * lane + k < NOFF) dbg[GC * lane + k] = t; }
        totals[k] = t;
        min_total = t < min_total ? t : min_total;
    });
    // ---- pass 2: first index whose mean int(total / N) is minimal (strict <, ref:332-337)
    const uint32_t m = __builtin_amdgcn_readfirstlane(wave_min_u32(min_total));
    const uint32_t bound = ((__umulhi(m, M) >> SH) + 1u) * (uint32_t)N;    // (min mean + 1) * N
    uint32_t cand = 0xFFFFFFFFu;
    static_for<0, GC>([&](auto kc) {
        constexpr int k = GC - 1 - decltype(kc)::value;                    // last to first: first wins
        cand = totals[k] < bound ? (uint32_t)k : cand;
    });
    cand = cand == 0xFFFFFFFFu ? cand : cand + (uint32_t)(GC * lane);
    cand = wave_min_u32(cand);
    return (int)__builtin_amdgcn_readfirstlane(cand);
}

// ---- phase A, the answer that needs no search (r5) -----------------------------------------------
// int(total(0) / N) == 0 -- the stream starts with the training sequence, sample for sample within a total of N -- is a
// mean no other offset can undercut (the means are sums of absolute values) at the first index there is: the
// reference's loop (ref:328-337: first index of the minimum, strict <) returns 0 whatever the other 4000 offsets
// hold.  That is every file Transmitter.save writes at 300 / 600 / 1200 baud ... (the training sequence starts at
// frame 0, ref:457; at 2400 baud the .wav writer's quirk leaves total(0) = 3276 * N and the search runs), and it is
// known as soon as the FIRST chunk(s) of the ring have landed: 64 lanes take one dword of the template span each,
// one wave reduction -- ~25 instructions instead of ~900, and the seven other chunks are not waited for.
template <int BF, int PRE = kRingChunks>
__device__ __forceinline__ bool clock_index_is_zero(FastRing& fr) {
    constexpr int N = 2 * BF, Q = BF / 4, H = BF / 2;
    constexpr uint32_t C = 65535u * (uint32_t)BF;
    constexpr int need = (2 * N - 1) >> 10;                   // chunk holding the last byte of the template span
    static_assert(need < 8, "the template span lies inside the sync window");
    fr.template wait_exact<PRE - 1 - need>(need);
    const int lane = fr.lane;
    int32_t a = 0;
#pragma unroll
    for (int it = 0; it < (BF + 63) / 64; it++) {
        const int m = lane + 64 * it;                         // dword m = samples 2m, 2m + 1
        const int mc = m < BF ? m : 0;
        const uint32_t w = *reinterpret_cast<const uint32_t*>(fr.ring + 4 * mc);
        uint32_t cf = 0;
#pragma unroll
        for (int half = 0; half < 2; half++) {
            const int j = 2 * mc + half;
            const bool hi = j < BF ? (((j / Q) & 1) == 0) : ((j - BF) < H);    // template 32767 here (ref:80-91)
            cf |= (hi ? 0xFFFFu : 0x0001u) << (16 * half);                     // sigma = -1 where it is
        }
        const int32_t v = dot2_i16(w, cf, 0);
        a += m < BF ? v : 0;
    }
    const int32_t sum = __builtin_amdgcn_readlane(wave_incl_scan_dpp(a), 63);
    return C + (uint32_t)sum < (uint32_t)N;                   // total(0) < N  <=>  int(total(0) / N) == 0
}

// ---- phase A, lane-wise form in steps (bit_frames 160) -----------------------------------
// A 300-baud lane window (72 + 320 samples) does not fit the register file, so the search runs
// in steps of 64 * GC offsets with GC = 24 per lane: a lane loads only the seven GC-sample
// sub-windows its deltas touch (7 x 3 ds_read_b128; all seven lags are multiples of 8
// samples, the 48-byte lane stride is bank-conflict free), forms the GC deltas with 7
// v_dot2c_i32_i16 each and their running sum; the total at a lane's first offset is the step's
// base plus the exclusive wave scan of the lane sums (one DPP scan per step), and the base of
// the next step is the base plus the scan's last element.  Only offset 0 needs a full
// correlation: 40 lanes take 8 samples each (8 divides the quarter symbol, so a lane's samples
// share one sign) and a wave reduction adds them up.
template <int BF, bool DEBUG = false, int PRE = kRingChunks>
__device__ __forceinline__ int recover_clock_index_lane_steps(FastRing& fr, uint32_t* dbg = nullptr,
                                                              unsigned long long* stamps = nullptr) {
    constexpr int N = 2 * BF, Q = BF / 4, H = BF / 2, NOFF = kSync - N;
    constexpr int GC = 24, STEP = 64 * GC, T = (NOFF + STEP - 1) / STEP;
    static_assert(2 * (STEP * (T - 1) + GC * 63 + N + GC) <= kRingBytes, "sub-windows outside the ring");
    constexpr uint32_t C = 65535u * (uint32_t)BF;
    // floor(m / N) = mul_hi(m, ceil(2^36 / N)) >> 4 while m * N < 2^36 (m <= 65535 * N: N <= 960);
    // longer templates use the float estimate + fix-up of div_exact (quotient < 2^16)
    constexpr bool MAGIC = 65535ull * N * N < (1ull << 36);
    constexpr uint32_t M = MAGIC ? (uint32_t)(((1ull << 36) + N - 1) / N) : 0u;
    static_assert((1ull << 36) / N < (1ull << 32), "magic divisor out of range");
    const int lane = fr.lane;
    using std::integral_constant;

    fr.template wait_exact<PRE - 8>(7);                       // chunks 0..7 (samples 0..4095) have landed
    if (stamps && lane == 0) stamps[2] = __builtin_amdgcn_s_memrealtime();
    // total(0) = C + sum_j sigma_j x[j] over the 2*BF template samples: dword m = samples 2m, 2m + 1,
    // lanes stride through the BF dwords; sigma = -1 where the template is 32767 (per sample: with an odd
    // quarter length a dword straddles a sign change)
    uint32_t base;
    {
        int32_t a = 0;
#pragma unroll
        for (int it = 0; it < (BF + 63) / 64; it++) {
            const int m = lane + 64 * it;
            const int mc = m < BF ? m : 0;
            const uint32_t w = *reinterpret_cast<const uint32_t*>(fr.ring + 4 * mc);
            uint32_t cf = 0;
#pragma unroll
            for (int half = 0; half < 2; half++) {
                const int j = 2 * mc + half;
                const bool hi = j < BF ? (((j / Q) & 1) == 0) : ((j - BF) < H);
                cf |= (hi ? 0xFFFFu : 0x0001u) << (16 * half);
            }
            const int32_t v = dot2_i16(w, cf, 0);
            a += m < BF ? v : 0;
        }
        const int32_t sum = __builtin_amdgcn_readlane(wave_incl_scan_dpp(a), 63);
        base = C + (uint32_t)sum;
    }
    uint32_t totals[T * GC];
    uint32_t min_total = 0xFFFFFFFFu;
    static_for<0, T>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        const uint8_t* src = fr.ring + 2 * (STEP * t + GC * lane);
        constexpr int lag[7] = {0, Q, 2 * Q, 3 * Q, BF, BF + H, N};
        constexpr int coef[7] = {1, -2, 2, -2, 2, -2, 1};
        // Sub-window e = the GC samples at lag[e] from the lane's first offset.  The lane base (48 bytes
        // per lane) is 16-byte aligned; a lag that is not a multiple of 8 samples is served by the ALIGNED
        // 64 bytes around it, re-aligned in registers by the compile-time shift (a 4-byte multiple is a
        // register renaming, 2 bytes cost one v_alignbyte per dword) -- misaligned ds_read_b128 execute on
        // gfx950 but several times slower (bit_frames 300 / 500: 78 -> 7x us per 4096 streams).
        uint32_t R[7][GC / 2];
        static_for<0, 7>([&](auto ec) {
            constexpr int e = decltype(ec)::value;
            constexpr int S = (2 * lag[e]) % 16;
            const uint8_t* p = src + 2 * lag[e] - S;
            if constexpr (S == 0) {
#pragma unroll
                for (int j = 0; j < GC / 8; j++) {
                    const u32x4 t4 = *reinterpret_cast<const u32x4*>(p + 16 * j);
                    R[e][4 * j] = t4[0]; R[e][4 * j + 1] = t4[1]; R[e][4 * j + 2] = t4[2]; R[e][4 * j + 3] = t4[3];
                }
            } else {
                uint32_t W[GC / 2 + 4];
#pragma unroll
                for (int j = 0; j < GC / 8 + 1; j++) {
                    const u32x4 t4 = *reinterpret_cast<const u32x4*>(p + 16 * j);
                    W[4 * j] = t4[0]; W[4 * j + 1] = t4[1]; W[4 * j + 2] = t4[2]; W[4 * j + 3] = t4[3];
                }
                realign_n<S, GC / 2 + 4, GC / 2>(W, R[e]);
            }
        });
        // run[k] = total(first + k + 1) - total(first)
        int32_t run[GC];
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
        const uint32_t first = base + (uint32_t)(incl - acc);          // total at this lane's first offset
        base += (uint32_t)__builtin_amdgcn_readlane(incl, 63);
#pragma unroll
        for (int k = 0; k < GC; k++) {
            uint32_t tot = k == 0 ? first : first + (uint32_t)run[k - 1];
            const int i = STEP * t + GC * lane + k;
            if constexpr (STEP * t + STEP > NOFF) tot = i < NOFF ? tot : 0xFFFFFFFFu;
            if constexpr (DEBUG) { if (i < NOFF) dbg[i] = tot; }
            totals[t * GC + k] = tot;
            min_total = tot < min_total ? tot : min_total;
        }
    });
    const uint32_t m = __builtin_amdgcn_readfirstlane(wave_min_u32(min_total));
    const uint32_t mean = MAGIC ? (__umulhi(m, M) >> 4) : div_exact(m, (uint32_t)N, 1.0f / (float)N);
    const uint32_t bound = (mean + 1u) * (uint32_t)N;                      // (min mean + 1) * N
    uint32_t cand = 0xFFFFFFFFu;
    static_for<0, T * GC>([&](auto kc) {
        constexpr int k = T * GC - 1 - decltype(kc)::value;                // last to first: first wins
        constexpr int i0 = STEP * (k / GC) + (k % GC);                     // offset of lane 0
        cand = totals[k] < bound ? (uint32_t)(i0 + GC * lane) : cand;
    });
    cand = wave_min_u32(cand);
    return (int)__builtin_amdgcn_readfirstlane(cand);
}

}  // namespace afsk
