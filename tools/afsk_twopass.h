// afsk_twopass.h -- the ROUND-1 two-pass demodulator, kept as the independent "v1" baseline of
// tools/kbench.hip (its outputs are what kbench compares every other variant with).  NOT part of the
// product: libafsk_amd.so does not contain it, and bench.py's kernel_source_hash does not cover it.
//
// Design: phase A builds a full 4096-entry int32 prefix array in LDS (7 lookups per sync offset),
// then phase B re-reads the stream from the clock index through a double-buffered, clock-index-
// aligned ring with the per-pass scalar state machine (rx_data: Hamming decode inside every pass).
// Include after afsk_demod_impl.h.
#pragma once

namespace afsk {

constexpr int kWaveLds = 16384;      // bytes of LDS per wave (= 4096 int32 prefix sums)

// Data part (ref:372-378) + Hamming decode + byte pack for the symbols [start, nv) of a pass.
__device__ __forceinline__ void rx_data(RxState& st, uint64_t bits, uint64_t amp_ok, int start,
                                        int nv, int lane, uint8_t* out_row, int out_stride) {
    if (st.phase != 1 || start < 0 || start >= nv) return;
    const uint64_t valid = nv >= 64 ? ~0ull : ((1ull << nv) - 1ull);
    bits &= valid;
    // take bits until the first symbol whose mean |x| < amp_end
    const uint64_t from = valid & ~((1ull << start) - 1ull);   // start < 64 here
    const uint64_t stop = from & ~amp_ok;
    int end = nv;
    if (stop) {
        end = __builtin_ctzll(stop);
        st.phase = 2;
    }
    const int n_new = end - start;
    if (n_new <= 0) return;
    uint64_t d = bits >> start;
    if (n_new < 64) d &= (1ull << n_new) - 1ull;
    st.nbits += n_new;
    // append to the pending coded bits; every 14 coded bits = 2 codewords = 1 byte
    const int np = st.npend;
    const uint64_t lo = (uint64_t)st.pend | (d << np);
    const uint64_t hi = np ? (d >> (64 - np)) : 0ull;
    const int total = np + n_new;       // <= 13 + 64
    const int nb = total / 14;          // <= 5
    if (lane < nb) {
        const int o = 14 * lane;        // <= 56
        uint32_t c = (uint32_t)(lo >> o);
        if (o > 50) c |= (uint32_t)(hi << (64 - o));
        c &= 0x3FFFu;
        const uint32_t byte = (hamming_nibble(c & 127u) << 4) | hamming_nibble(c >> 7);  // ref:393-399
        const int pos = st.nbytes + lane;
        if (pos < out_stride) out_row[pos] = (uint8_t)byte;
    }
    {   // soft output: corrected codewords among the 2*nb just decoded
        uint32_t cc = 0;
        if (lane < nb) {
            const int o = 14 * lane;
            uint32_t c = (uint32_t)(lo >> o);
            if (o > 50) c |= (uint32_t)(hi << (64 - o));
            cc = (hamming_syndrome(c & 127u) != 0) + (hamming_syndrome((c >> 7) & 127u) != 0);
        }
        st.corrected += (int32_t)__popcll(__ballot(cc >= 1)) + (int32_t)__popcll(__ballot(cc >= 2));
    }
    st.nbytes += nb;
    const int used = 14 * nb;           // <= 70
    const int rem = total - used;       // < 14
    uint64_t rest;
    if (used == 0) rest = lo;
    else if (used < 64) rest = (lo >> used) | (hi << (64 - used));
    else if (used == 64) rest = hi;
    else rest = hi >> (used - 64);
    st.pend = (uint32_t)rest & ((1u << rem) - 1u);
    st.npend = rem;
}

// bits / amp_ok: bit j = decision of symbol k0+j (only j < nv meaningful).
__device__ __forceinline__ void rx_consume(RxState& st, uint64_t bits, uint64_t amp_ok, int nv,
                                           int k0, int lane, uint8_t* out_row, int out_stride) {
    const int start = rx_training(st, bits, nv, k0);
    rx_data(st, bits, amp_ok, start, nv, lane, out_row, out_stride);
}

// ------------------------------------------------------------------- phase A
// ref:322-339.  Returns the clock index (wave-uniform).  P = 4096 int32 in LDS.
__device__ __forceinline__ int recover_clock_index(const int16_t* xs, int32_t len, int bf,
                                                   int32_t* P, int lane) {
    auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)xs, 0, len * 2, 0x00020000);
    u32x4 v[8];
#pragma unroll
    for (int r = 0; r < 8; r++)
        v[r] = __builtin_bit_cast(
            u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (512 * r + 8 * lane) * 2, 0, 0));
    int32_t c[8][8];   // exclusive prefix inside the lane's 8 samples
    int32_t tot[8];
#pragma unroll
    for (int r = 0; r < 8; r++) {
        int32_t run = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int32_t w = (int32_t)v[r][j];
            c[r][2 * j] = run;
            run += (w << 16) >> 16;
            c[r][2 * j + 1] = run;
            run += w >> 16;
        }
        tot[r] = run;
    }
    int32_t incl[8];
#pragma unroll
    for (int r = 0; r < 8; r++) incl[r] = wave_incl_scan_dpp(tot[r]);
    int32_t carry = 0;
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const int32_t base = carry + incl[r] - tot[r];
        int32_t* dst = P + 512 * r + 8 * lane;
        u32x4 a = {(uint32_t)(base + c[r][0]), (uint32_t)(base + c[r][1]),
                   (uint32_t)(base + c[r][2]), (uint32_t)(base + c[r][3])};
        u32x4 b = {(uint32_t)(base + c[r][4]), (uint32_t)(base + c[r][5]),
                   (uint32_t)(base + c[r][6]), (uint32_t)(base + c[r][7])};
        *reinterpret_cast<u32x4*>(dst) = a;
        *reinterpret_cast<u32x4*>(dst + 4) = b;
        carry += __builtin_amdgcn_readlane(incl[r], 63);
    }
    // Same wave wrote and reads P: LDS ops of one wave complete in order.
    wave_lds_sync();
    const int q = bf >> 2, h = bf >> 1, n = 2 * bf;
    const int n_off = kSync - n;                 // ref:327
    const uint32_t C = 65535u * (uint32_t)bf;    // 32767*bf + 32768*bf
    const float rcp_n = 1.0f / (float)n;
    uint32_t best = 0xFFFFFFFFu;
    for (int i0 = 0; i0 < n_off; i0 += 64) {
        const int i = i0 + lane;
        const bool ok = i < n_off;
        const int ic = ok ? i : n_off - 1;
        const int32_t* p = P + ic;
        // training cycle = mark(hi q, lo q, hi q, lo q) + space(hi h, lo h), ref:80-91
        const int32_t t = p[0] + p[n] +
                          2 * (p[2 * q] + p[bf] - p[q] - p[3 * q] - p[bf + h]);
        const uint32_t total = C + (uint32_t)t;                 // sum |tc[j] - x[i+j]|
        const uint32_t mean = div_exact(total, (uint32_t)n, rcp_n);   // ref:107
        const uint32_t key = (mean << 12) | (uint32_t)ic;
        if (ok && key < best) best = key;        // strict <, first minimum: ref:332-337
    }
    best = wave_min_u32(best);
    return (int)(__builtin_amdgcn_readfirstlane(best) & 4095u);
}

// Specialised symbol loop: BF samples per symbol, M lanes per symbol (each lane a
// contiguous piece of PL = BF/M samples).  Requires M == 1, or a piece that lies
// inside one quarter symbol (templates are then one constant per lane).
template <int BF, int M, int FLAGS = 0>
__device__ __forceinline__ void demod_symbols(const int16_t* xs, int32_t len, int ci,
                                              int32_t amp_end, uint8_t* lds, int lane,
                                              RxState& st, uint8_t* out_row, int out_stride,
                                              int32_t& n_sym_out) {
    constexpr int PL = BF / M;                   // samples per lane piece
    constexpr int Q = BF / 4, H = BF / 2;
    static_assert(BF % 4 == 0 && PL % 2 == 0, "piece must be whole dwords");
    static_assert(M == 1 || (Q % PL == 0), "piece must lie inside one quarter");
    constexpr int PIECE_B = PL * 2;              // bytes per lane piece
    constexpr int PASS_B = 64 * PIECE_B;         // bytes per 64-lane pass
    constexpr int U = (PASS_B % 1024 == 0) ? 1 : ((2 * PASS_B) % 1024 == 0 ? 2 : 4);
    constexpr int RB = U * PASS_B;               // bytes per DMA round
    static_assert(RB % 1024 == 0, "round must be whole 1 KiB DMA instructions");
    constexpr int NCH = RB / 1024;               // DMA instructions per round
    constexpr int NSLOT = kWaveLds / RB;         // ring depth
    static_assert(NSLOT >= 2, "ring needs two slots");
    constexpr int SPP = 64 / M;                  // symbols per pass
    constexpr int SPR = SPP * U;                 // symbols per round

    const int32_t rel_len = len - ci;                          // samples from the clock index
    const int32_t K = (rel_len - BF + BF - 1) / BF;            // symbols with i < len - bf (ref:362,372)
    n_sym_out = K;
    const int32_t NR = (K + SPR - 1) / SPR;
    auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(xs + ci), 0, rel_len * 2, 0x00020000);
    const uint32_t amp_thr =
        (uint32_t)(amp_end < 0 ? 0 : (amp_end > 40000 ? 40000 : amp_end)) * (uint32_t)BF;

    auto issue_round = [&](int r) {
        uint8_t* slot = lds + (r % NSLOT) * RB;
#pragma unroll
        for (int c = 0; c < NCH; c++)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, AFSK_LDS(slot + c * 1024), 16,
                                                     lane * 16 + c * 1024, r * RB, 0, 0);
    };

#pragma unroll
    for (int r = 0; r < NSLOT - 1; r++)
        if (r < NR) issue_round(r);

    // per-lane constant templates when M > 1
    const int piece = lane % M;
    const int quarter = (piece * PL) / Q;
    const uint32_t tm_lane = (quarter & 1) ? 0u : 0xFFFFFFFFu;
    const uint32_t ts_lane = quarter < 2 ? 0xFFFFFFFFu : 0u;

    for (int r = 0; r < NR; r++) {
        if (r + NSLOT - 1 < NR) {
            issue_round(r + NSLOT - 1);
            wait_vmcnt<NCH*(NSLOT - 1)>();
        } else {
            wait_vmcnt<0>();
        }
        const uint8_t* slot = lds + (r % NSLOT) * RB;
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int k0 = r * SPR + u * SPP;
            if (k0 >= K) break;
            const uint8_t* src = slot + u * PASS_B + lane * PIECE_B;
            uint32_t w[PL / 2];
            if constexpr (PIECE_B % 16 == 0) {
#pragma unroll
                for (int j = 0; j < PIECE_B / 16; j++) {
                    u32x4 t = *reinterpret_cast<const u32x4*>(src + 16 * j);
                    w[4 * j] = t[0]; w[4 * j + 1] = t[1]; w[4 * j + 2] = t[2]; w[4 * j + 3] = t[3];
                }
            } else if constexpr (PIECE_B % 8 == 0) {
#pragma unroll
                for (int j = 0; j < PIECE_B / 8; j++) {
                    u32x2 t = *reinterpret_cast<const u32x2*>(src + 8 * j);
                    w[2 * j] = t[0]; w[2 * j + 1] = t[1];
                }
            } else {
#pragma unroll
                for (int j = 0; j < PIECE_B / 4; j++)
                    w[j] = *reinterpret_cast<const uint32_t*>(src + 4 * j);
            }
            uint32_t mark = 0, space = 0, amp = 0;
            if constexpr (FLAGS & 2) {   // diagnostic: keep the loads live, skip the arithmetic
#pragma unroll
                for (int d = 0; d < PL / 2; d++) amp |= w[d];
                mark = amp & 1; space = 1; amp = 0x7fffffff;
            } else
#pragma unroll
            for (int d = 0; d < PL / 2; d++) {
                const uint32_t x = w[d];
                const uint32_t lim = limit_pair_biased(x);                      // ref:344
                uint32_t tm, ts;
                if constexpr (M == 1) {
                    tm = mark_half(2 * d, Q) | (mark_half(2 * d + 1, Q) << 16);
                    ts = space_half(2 * d, H) | (space_half(2 * d + 1, H) << 16);
                } else {
                    tm = tm_lane;
                    ts = ts_lane;
                }
                mark = __builtin_amdgcn_sad_u16(lim, tm, mark);                 // ref:346
                space = __builtin_amdgcn_sad_u16(lim, ts, space);               // ref:347
                amp = __builtin_amdgcn_sad_u16(x ^ kBias, kBias, amp);          // ref:94-98
            }
            if constexpr (M > 1) {
#pragma unroll
                for (int s = 1; s < M; s <<= 1) {
                    mark += (uint32_t)__shfl_xor((int)mark, s, 64);
                    space += (uint32_t)__shfl_xor((int)space, s, 64);
                    amp += (uint32_t)__shfl_xor((int)amp, s, 64);
                }
            }
            const bool bit = (mark / (uint32_t)BF) < (space / (uint32_t)BF);    // ref:348-351
            const bool loud = amp >= amp_thr;   // !(int(sum/bf) < amp_end), ref:375
            uint64_t bmask, amask;
            if constexpr (M == 1) {
                bmask = __ballot(bit);
                amask = __ballot(loud);
            } else {
                // lane j < SPP picks up symbol j's decision from lane j*M
                const int srcl = (lane * M) & 63;
                const int pk = __shfl((int)bit | ((int)loud << 1), srcl, 64);
                bmask = __ballot((pk & 1) && lane < SPP);
                amask = __ballot((pk & 2) && lane < SPP);
            }
            const int nv = (K - k0) < SPP ? (K - k0) : SPP;
            rx_consume(st, bmask, amask, nv, k0, lane, out_row, out_stride);
            if (st.phase == 2) break;
        }
        if (st.phase == 2) break;
    }
    wait_vmcnt<0>();   // drain DMA still in flight before the LDS region is released
}

__device__ __forceinline__ void wait_vmcnt_upto8(int n) {
    switch (n) {
        case 0: wait_vmcnt<0>(); break;   case 1: wait_vmcnt<1>(); break;
        case 2: wait_vmcnt<2>(); break;   case 3: wait_vmcnt<3>(); break;
        case 4: wait_vmcnt<4>(); break;   case 5: wait_vmcnt<5>(); break;
        case 6: wait_vmcnt<6>(); break;   case 7: wait_vmcnt<7>(); break;
        default: wait_vmcnt<8>(); break;
    }
}

// Generic fallback: any bf (multiple of 4, 2*bf < 4096), one lane per symbol.  Correctness
// path for unusual baud rates: two 8 KiB LDS slots (the DMA of round r+1 overlaps the
// arithmetic of round r), four samples per step with the packed limiter and v_sad_u16; the
// mark/space templates of a step are the same for every lane (all symbols of a round are at
// the same phase), so the quarter bookkeeping stays on the scalar unit.
__device__ __forceinline__ void demod_symbols_generic(const int16_t* xs, int32_t len, int ci,
                                                      int bf, int32_t amp_end, uint8_t* lds,
                                                      int lane, RxState& st, uint8_t* out_row,
                                                      int out_stride, int32_t& n_sym_out,
                                                      int32_t* margins = nullptr,
                                                      int32_t margin_stride = 0) {
    constexpr int kSlot = kWaveLds / 2;
    const int q = bf >> 2, h = bf >> 1;
    const int sym_b = bf * 2;                          // bytes per symbol, a multiple of 8
    int spr = kSlot / sym_b;                           // symbols per round: 2 .. 64
    if (spr > 64) spr = 64;
    const int rb = spr * sym_b;
    const int nch = (rb + 1023) >> 10;                 // 1 KiB DMA instructions per round, <= 8
    const int32_t rel_len = len - ci;
    const int32_t K = (rel_len - bf + bf - 1) / bf;    // symbols with i < len - bf (ref:362,372)
    n_sym_out = K;
    const int32_t NR = (K + spr - 1) / spr;
    auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(xs + ci), 0, rel_len * 2, 0x00020000);
    const uint32_t amp_thr =
        (uint32_t)(amp_end < 0 ? 0 : (amp_end > 40000 ? 40000 : amp_end)) * (uint32_t)bf;
    const float rcp_bf = 1.0f / (float)bf;
    auto issue_round = [&](int r) {
        uint8_t* slot = lds + (r & 1) * kSlot;
        for (int c = 0; c < nch; c++) {
            if (c * 1024 + lane * 16 < rb)             // lane 0 always issues: vmcnt counts nch per round
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, AFSK_LDS(slot + c * 1024), 16,
                                                         lane * 16 + c * 1024, r * rb, 0, 0);
        }
    };
    if (NR > 0) issue_round(0);
    for (int r = 0; r < NR; r++) {
        if (r + 1 < NR) {
            issue_round(r + 1);
            wait_vmcnt_upto8(nch);                     // round r has landed, round r+1 may be in flight
        } else {
            wait_vmcnt<0>();
        }
        const uint8_t* src = lds + (r & 1) * kSlot + (lane < spr ? lane : 0) * sym_b;
        uint32_t mark = 0, space = 0, amp = 0;
        int qi = 0, nb = q;                            // quarter of the current phase, next boundary (uniform)
        for (int it = 0; it < q; it++) {               // 4 samples per step: phases 4*it .. 4*it+3
            const u32x2 w = *reinterpret_cast<const u32x2*>(src + 8 * it);
            uint32_t tmh[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (4 * it + j == nb) { qi++; nb += q; }
                tmh[j] = (qi & 1) ? 0x0000u : 0xFFFFu;                         // hi on quarters 0, 2 (ref:80-85)
            }
            const uint32_t ts = 4 * it < h ? 0xFFFFFFFFu : 0u;                 // hi on the first half (ref:68-77; h % 2 == 0)
            const uint32_t l0 = limit_pair_biased(w[0]), l1 = limit_pair_biased(w[1]);   // ref:344
            mark = __builtin_amdgcn_sad_u16(l0, tmh[0] | (tmh[1] << 16), mark);          // ref:346
            mark = __builtin_amdgcn_sad_u16(l1, tmh[2] | (tmh[3] << 16), mark);
            space = __builtin_amdgcn_sad_u16(l0, ts, space);                             // ref:347
            space = __builtin_amdgcn_sad_u16(l1, (4 * it + 2 < h) ? 0xFFFFFFFFu : 0u, space);
            amp = __builtin_amdgcn_sad_u16(w[0] ^ kBias, kBias, amp);                    // ref:94-98
            amp = __builtin_amdgcn_sad_u16(w[1] ^ kBias, kBias, amp);
        }
        const uint32_t md = div_exact(mark, (uint32_t)bf, rcp_bf), sd = div_exact(space, (uint32_t)bf, rcp_bf);
        const bool bit = md < sd;                                              // ref:348-351
        const bool loud = amp >= amp_thr;                                      // ref:375
        const uint64_t bmask = __ballot(bit && lane < spr);
        const uint64_t amask = __ballot(loud && lane < spr);
        const int k0 = r * spr;
        if (margins && lane < spr && k0 + lane < K && k0 + lane < margin_stride)
            margins[k0 + lane] = (int32_t)sd - (int32_t)md;
        const int nv = (K - k0) < spr ? (K - k0) : spr;
        // every lane's reads of this slot have returned before round r+2 overwrites it
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        rx_consume(st, bmask, amask, nv, k0, lane, out_row, out_stride);
        if (st.phase == 2) break;
    }
    wait_vmcnt<0>();
}

template <int FLAGS>
__global__ __launch_bounds__(64 * kWavesPerBlock) void demod_twopass_kernel_t(DemodArgs a) {
    __shared__ __attribute__((aligned(16))) uint8_t lds_all[kWavesPerBlock * kWaveLds];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int s = blockIdx.x * kWavesPerBlock + wave;
    if (s >= a.n_streams) return;
    const int32_t len = a.stream_len[s];
    const int bf = a.bit_frames[s];
    uint8_t* lds = lds_all + wave * kWaveLds;
    if (!bit_frames_valid(bf)) { store_refusal(a, s, lane, 3); return; }
    if (len < kSync) { store_refusal(a, s, lane, 1); return; }
    const int16_t* xs = a.samples + a.stream_offset[s];
    uint8_t* out_row = a.out_bytes + (int64_t)s * a.out_stride;
    RxState st;
    st.phase = 0; st.hist = 0; st.nbits = 0; st.nbytes = 0; st.term_sym = -1; st.pend = 0;
    st.npend = 0; st.corrected = 0;
    int32_t n_sym = 0;
    int ci = 0;
    if constexpr (!(FLAGS & kFlagSkipSync))
        ci = recover_clock_index(xs, len, bf, reinterpret_cast<int32_t*>(lds), lane);
    // phase B reuses the prefix-sum region: all LDS reads of phase A have returned
    // (their values were consumed), so the DMA writes below cannot overtake them.
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    switch (bf) {
        case 40:  demod_symbols<40, 1, FLAGS>(xs, len, ci, a.amp_end, lds, lane, st, out_row, a.out_stride, n_sym); break;
        case 20:  demod_symbols<20, 1, FLAGS>(xs, len, ci, a.amp_end, lds, lane, st, out_row, a.out_stride, n_sym); break;
        case 160: demod_symbols<160, 4, FLAGS>(xs, len, ci, a.amp_end, lds, lane, st, out_row, a.out_stride, n_sym); break;
        default:  demod_symbols_generic(xs, len, ci, bf, a.amp_end, lds, lane, st, out_row, a.out_stride, n_sym); break;
    }
    store_result(a, s, lane, st, ci, n_sym, bf);
}

}  // namespace afsk
