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
//   phase B   ref:342-351: every lane owns an 80-byte piece (40 samples: one
//             1200-baud symbol, two 2400-baud symbols, half a 600-baud or a quarter of a
//             300-baud symbol) at ring byte (2*ci + 5120*r + 80*lane) mod 16 KiB, read as five
//             aligned ds_read_b128, or six re-aligned in registers by the wave-uniform shift
//             (2*ci) & 15 (v_alignbyte).  After the reads of round r the five chunks
//             it consumed are refilled immediately (11 KiB stay in flight).
//   phase C   ref:361-378, 145-163, 393-399: terminator scan and squelch stop on wave-uniform
//             ballot masks; the Hamming decode + byte pack is deferred and vectorised.
#pragma once
#include <type_traits>

namespace afsk {

constexpr int kRingBytes = 16384;
constexpr int kRingChunks = 16;
// LDS of a wave behind the ring
constexpr int kMirrorBytes = 256;                              // copy of ring bytes 0..255 right behind the ring: a lane's
                                                               // piece may run linearly past the ring end (wm_rounds)
constexpr int kBitBufOffset = kRingBytes + kMirrorBytes;       // phase C's 64-word bit buffer behind the mirror
constexpr int kBitBufBytes = 512;
constexpr int kWarmDummyOffset = kBitBufOffset + kBitBufBytes;  // 256 bytes the warming requests may scribble on
constexpr int kProbeOffset = kWarmDummyOffset + 256;            // 64 x 16 bytes: the tail-hint probes land here
constexpr int kHintStashOffset = kProbeOffset + 1024;           // 16 bytes: probe spacing, parked here instead of in scalar registers

// L2 warming behind the ring start (r2).  While a wave computes phase A its 16 ring chunks have
// landed and it has nothing in flight -- LDS caps the ring at 16 KiB.  Right behind the 16 chunk
// requests the wave therefore asks for one dword of every 64 bytes of stream bytes 16 KiB .. 24 KiB
// (two LDS-DMA instructions into a 256-byte dummy area, default cache policy): the lines are
// fetched HBM -> L2 during phase A and the real requests for chunks 16..23 then hit in L2.
// Worth 2.7-4 % in steady state (16384+ streams); at 4096 streams, where all waves of a generation
// start together, it costs 1-2 %, so it is only armed for launches of kWarmMinStreams or more.
// A run-ahead kept up for the whole stream (two more warming requests per round) is 10 % SLOWER:
// every line is then requested twice and the request path, not HBM, becomes the limit.
constexpr int kWarmOps = 2;
constexpr int kWarmMinStreams = 8192;
// Tail hint (r2, for launches of kHintMinStreams or more).
// A stream ends in silence (4800 zero samples behind every Transmitter frame, ref:468) that the
// reference never reads -- it stops at the first quiet symbol -- but a prefetching reader requests it
// long before it can know: ~10 KiB are in flight when the squelch fires, i.e. the whole 9.6 KB tail.
// So, once phase A is done, the wave requests kProbes (one per lane) 16-byte probes, each the last 8 samples of a round
// (of every m-th round, so that kProbes of them cover the stream; one LDS-DMA instruction, 2 KiB of
// HBM traffic), and when they have landed it looks for the LAST probe that is loud by the squelch's
// own measure (sum of the 8 |x| >= 8 * amp_end; 8 samples, so that noise in the tail -- config #4 --
// rarely looks loud): the signal then ends inside the round group closed by the
// next probe, and chunks behind that group are not requested AHEAD OF NEED any more.  This is a prefetch policy only: a round that needs a chunk which was held back requests it
// on the spot (and drops the hint), so results cannot change -- e.g. a weak signal below amp_end
// with no loud probe at all still decodes, one demand fetch later.
constexpr int kProbes = 64;               // one per lane; probes beyond the stream end cost nothing (range-checked)
constexpr int kHintMinStreams = 6144;     // mixed-baud kernel: -2.6 % at 6144 streams, -0.7 ... +1.3 % at 4096 and below
constexpr int kHintMinStreamsGrouped = 4096;   // the same kernel walking a rate-SORTED stream list (grouped dispatch): the
                                               // +1.3 % at 4096 was measured in stream order (r5)
// uniform kernels of bit_frames 4 / 8 (large-launch form, hint and warming alike): 12000 baud gains from 8192
// streams on (0.65 -> 0.68; 16384: 0.65 -> 0.71; 32768: 0.72 -> 0.77), 6000 baud loses 2 % at 8192 / 12288 and gains
// from 16384 on (0.69 -> 0.71; 32768: 0.69 -> 0.76) -- profiles/r4_exp4_hint_short.txt
constexpr int kHintMinStreamsShort4 = 8192;
constexpr int kHintMinStreamsShort8 = 16384;
constexpr int kHintMinStreamsUniform = 4096;   // uniform kernels (no scalar-register pressure): -0.8 ... -1.6 % at 4096 streams and 1.02 x
                                               // instead of 1.11 x the algorithmic bytes fetched; neutral at 2048

// Squelch amplitude (ref:94-98, ref:375) without a bias instruction per dword (r5).  v_sad_u16 of the RAW packed
// pair against 0x8000 per half gives, per sample, 32768 - |x|: a non-negative sample x reads as x (32768 - x), a
// negative one as 65536 + x (minus 32768: 32768 - |x|), and -32768 gives 0 = 32768 - abs(-32768) like the
// reference's Python abs.  So the "quiet sum" q of n samples is 32768 n - sum|x|, and
//     sum|x| >= thr   <=>   q <= 32768 n - thr      (signed: thr may exceed 32768 n, then nothing is ever loud).
// (r1-r4 formed |x| itself: v_xor with 0x80008000, then the same v_sad_u16 -- twice the instructions.)
__device__ __forceinline__ uint32_t quiet_sad(uint32_t x, uint32_t acc) { return __builtin_amdgcn_sad_u16(x, kBias, acc); }
__device__ __forceinline__ bool loud_enough(uint32_t quiet, uint32_t n_samples, uint32_t amp_thr) {
    return (int32_t)quiet <= (int32_t)(32768u * n_samples) - (int32_t)amp_thr;
}

// Tail hint, second level (r5): with round-spaced probes alone a wave fetches up to one round past the end of the signal
// -- half a round on average, 1.07 x the algorithmic bytes at 4000 baud, 6 KiB rounds (PMC).  Once the first level has
// found the probe interval in which the signal ends, EIGHT more probes inside that interval (an eighth of it apart:
// 0.6 - 1 KiB) narrow the limit down to a chunk; the round that then reaches past the limit is decoded from what has
// been requested first (FastRing::holding_wait).  -1.6 ... -4.9 % where the one-level hint happened to waste most
// (2000 / 1000 / 800 / 500 / 400 / 375 / 96 baud), neutral where the bench's payload sizes end near a round boundary
// anyway (1200 / 2400 / 300 baud ...: profiles/r5_exp16_two_level_hint.txt, r5_exp17_*).  Measured alternative
// (r5_exp14/15): 64 probes 1.5 KiB apart from the start cost 47 more requests and 2 - 3 KB of traffic per stream --
// +2 ... 4 % for 5 KiB rounds, a wash for 6 KiB ones.  AFSK_REFINE_FROM: smallest round that takes the second level.
#ifndef AFSK_REFINE_FROM
#define AFSK_REFINE_FROM 3072
#endif
__host__ __device__ constexpr bool fine_probes(int round_bytes) { return round_bytes >= (AFSK_REFINE_FROM); }

struct FastRing {
    __amdgpu_buffer_rsrc_t rsrc;   // whole stream: base = sample 0, num_records = 2*len
    uint8_t* ring;                 // wave-uniform LDS base of the 16 KiB ring
    int lane;
    int next;                      // next chunk id to issue
    int warm_ops = 0;              // warming requests issued between chunk 15 and chunk 16 (0 or kWarmOps)

    // Requests complete in issue order, so chunk `need` has landed once at most as many requests as
    // were issued after it are outstanding; FIXED = that count without the warming requests.
    // (r5: always the immediate.  The warming / probe requests sit between chunk 15 and chunk 16 in issue order, so
    // while need < 16 this waits for up to warm_ops requests more than it has to -- requests that were issued
    // one or two rounds earlier and have landed -- instead of running a scalar test and a switch in EVERY round:
    // every instruction, scalar ones too, costs a wave four cycles of its issue slot.)
    template <int FIXED>
    __device__ __forceinline__ void wait_fixed(int /*need*/) {
        wait_vmcnt<FIXED>();
    }
    // the exact form, for the one wait per stream in front of phase A (two chunks more would delay its start)
    template <int FIXED>
    __device__ __forceinline__ void wait_exact(int need) {
        if (warm_ops != 0 && need < kRingChunks) wait_vmcnt_dyn(FIXED + warm_ops);
        else wait_vmcnt<FIXED>();
    }

    template <int AUX = 0>
    __device__ __forceinline__ void issue(int c) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, AFSK_LDS(ring + (c & (kRingChunks - 1)) * 1024),
                                                 16, lane * 16, c * 1024, 0, AUX);
    }

    // ---- tail hint (see kProbes) ----  (state kept small: the round loops are short of scalar registers)
    int hint_state = 0;            // bit 0: probes requested for this stream, bit 1: evaluated, bit 2: the hint is
                                   // holding chunks back (round loops with a fixed schedule switch to the
                                   // dynamic one), bits 3..: misses
    int hint_lim = 0x7fffffff;     // chunks at or above this index are not requested ahead of need
    int eval_need = 0x7fffffff;    // the probes are evaluated in the first round whose last chunk is >= this (one
                                   // compare per round: request_probes arms it with kRingChunks, eval_probes disarms it)

    __device__ __forceinline__ bool hint_armed() const { return (hint_state & 1) != 0; }
    __device__ __forceinline__ bool hint_holding() const { return (hint_state & 4) != 0; }
    // a loop that requests a fixed number of chunks per round calls this before doing so: true (and
    // sticky) once that request would cross the hint
    __device__ __forceinline__ bool hint_takes_over(int chunks_per_round) {
        if (next + chunks_per_round > hint_lim) hint_state |= 4;
        return (hint_state & 4) != 0;
    }

    // Requested after phase A and before chunk 16, so the probes do not compete with the wave's first
    // 16 KiB and count like the warming requests ("between chunk 15 and chunk 16") in the waits.
    // Probe j is the last 16 bytes (8 samples) below stream byte base + (j + 1) * step, step = m rounds with m chosen
    // so that kProbes of them cover the stream: a probe sits at the END OF A ROUND, and if it is quiet
    // and the one before it loud, the signal ends inside the rounds between them and the last chunk the
    // decoder can need is the one holding that very dword.
    __device__ __forceinline__ void request_probes(uint32_t stream_bytes, int base, int round_bytes) {
        const uint32_t span = stream_bytes > (uint32_t)base ? stream_bytes - (uint32_t)base : 0u;
        const uint32_t rounds = span / (uint32_t)round_bytes + 1u;
        const int step = (int)(((rounds + kProbes - 1) / kProbes) * (uint32_t)round_bytes);
        if (lane == 0) *reinterpret_cast<int*>(ring + kHintStashOffset) = step;
        const uint32_t po = (uint32_t)base + (uint32_t)((lane & (kProbes - 1)) + 1) * (uint32_t)step;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, AFSK_LDS(ring + kProbeOffset), 16, (int)(po - 16u), 0, 0, 0);
        warm_ops += 1;
        hint_state = 1;
        eval_need = kRingChunks;
    }
    // chunk `need` has landed (dynamic form of wait_fixed: the number of requests behind it varies
    // once chunks are held back)
    __device__ __forceinline__ void wait_landed(int need) {
        wait_vmcnt_dyn(next - 1 - need + (need < kRingChunks ? warm_ops : 0));
    }
    // a round needs chunk `need`: request whatever of it was held back.  One such miss is normal (noise
    // can push the stop one symbol into the next round); a second one means the hint is wrong: drop it.
    template <int AUX>
    __device__ __forceinline__ void fetch_through(int need) {
        if (next > need) return;
        hint_state += 8;
        hint_lim = hint_state >= 16 ? 0x7fffffff : need + 1;
        while (next <= need) {
            issue<AUX>(next);
            next++;
        }
    }
    // Holding mode, top of a round that reads stream bytes up to `last` (inclusive) and starts at symbol k0 (symbol 0 at
    // byte `base`, sym_bytes each; K symbols in the stream).  Returns how many symbols the round may use:
    //   K       everything it reads has landed -- requested earlier, or (nothing of the round available: the r4
    //           behaviour) fetched through now, or lying past the stream's last symbol;
    //   < K     PARTIAL: the round reaches past what has been requested.  It is decoded from the symbols that lie wholly
    //           below the requested bytes first: the squelch stop is almost always among them (that is what the probes
    //           said), and then nothing more is ever fetched.  If it is not, the caller restores its state, fetches the
    //           rest (fetch_through: a miss) and runs the round again -- results cannot depend on the hint.
    template <int AUX>
    __device__ __forceinline__ int32_t holding_wait(int last, int32_t K, int32_t k0, int base, int sym_bytes, bool& partial) {
        const int need = last >> 10;
        partial = false;
        if (next <= need) {
            const int32_t kp = (int32_t)(((uint32_t)next * 1024u - (uint32_t)base) / (uint32_t)sym_bytes);
            if (kp >= K) {                                   // only bytes behind the last symbol are missing
                wait_landed(next - 1);
                return K;
            }
            if (kp > k0) {
                partial = true;
                wait_landed(next - 1);
                return kp;
            }
            fetch_through<AUX>(need);
        }
        wait_landed(need);
        return K;
    }
    // request every chunk below lim (that the hint allows)
    template <int AUX, bool HINTED = true>
    __device__ __forceinline__ void top_up(int lim) {
        if constexpr (HINTED) lim = lim < hint_lim ? lim : hint_lim;
        while (next < lim) {
            issue<AUX>(next);
            next++;
        }
    }
    // ---- rounds that are not whole chunks (wm_rounds / gp_rounds), the common case in two tests (r5) ----
    // `pos` = first stream byte of the round, RB = bytes from there to the last byte it reads, inclusive.  While the
    // refill runs at the watermark -- next == (pos >> 10) + 16: every chunk below the round's first byte has been
    // requested again and nothing is held back -- at least 15 - CMAX requests were issued behind the chunk of the
    // round's last byte (CMAX = the most chunk boundaries RB bytes can cross), so that immediate is a sufficient
    // wait (one chunk more than necessary in the rounds that cross fewer).  Otherwise: the exact, dynamic form.
    template <int AUX, int RB, bool HINTED>
    __device__ __forceinline__ int32_t wait_round(int pos, int32_t K, int32_t k0, int base, int sym_bytes, bool& partial) {
        constexpr int CMAX = (RB + 1023) >> 10;
        static_assert(CMAX < kRingChunks - 1, "round too large for the ring");
        partial = false;
        if (next == (pos >> 10) + kRingChunks) {
            wait_vmcnt<kRingChunks - 1 - CMAX>();
            return K;
        }
        if constexpr (HINTED) return holding_wait<AUX>(pos + RB - 1, K, k0, base, sym_bytes, partial);
        wait_landed((pos + RB - 1) >> 10);
        return K;
    }
    // refill behind a round of RBYTES bytes: every chunk wholly below the next round's first byte.  At the
    // watermark that is RBYTES >> 10 chunks or one more: straight-line requests instead of a loop.
    template <int AUX, int RBYTES, bool HINTED>
    __device__ __forceinline__ void refill_round(int pos) {
        constexpr int CMIN = RBYTES >> 10;
        const int lim = ((pos + RBYTES) >> 10) + kRingChunks;
        if (next == (pos >> 10) + kRingChunks && (!HINTED || lim <= hint_lim)) {
#pragma unroll
            for (int j = 0; j < CMIN; j++) issue<AUX>(next + j);
            next += CMIN;
            if (next < lim) { issue<AUX>(next); next++; }
        } else {
            top_up<AUX, HINTED>(lim);
        }
    }
    // once a chunk >= 16 has landed the probes have too: hold back everything behind the round group
    // whose closing probe is the first quiet one after the last loud one (amp1 = the squelch threshold
    // per sample, 0 = nothing is ever quiet; base as given to request_probes; extra = bytes a round
    // reads past its end when re-aligning).  REFINE: the second level (see fine_probes) -- eight probes inside that
    // group, requested here and evaluated once a chunk requested after them has landed; `margin` bytes (one symbol)
    // are added to the refined limit: the squelch stops at the first quiet SYMBOL, which may start behind a quiet probe.
    template <bool REFINE = false>
    __device__ __forceinline__ void eval_probes(int need, uint32_t amp1, int base, int extra, int margin = 0) {
        if (need < eval_need) return;
        eval_need = 0x7fffffff;
        wave_lds_sync();
        const u32x4 pv = *reinterpret_cast<const u32x4*>(ring + kProbeOffset + 16 * lane);
        uint32_t q8 = 0;                                                                   // 8 * 32768 - (|x0| + ... + |x7|)
#pragma unroll
        for (int j = 0; j < 4; j++) q8 = quiet_sad(pv[j], q8);
        const uint64_t loud = __ballot(loud_enough(q8, 8u, 8u * amp1));
        if constexpr (REFINE) {
            if (hint_state & 2) {                             // ---- second level: lanes 0..7 hold the sub-probes
                if (hint_state >= 8) return;                  // a miss has moved the limit since: leave it alone
                const int base2 = __builtin_amdgcn_readfirstlane(*reinterpret_cast<const int*>(ring + kHintStashOffset + 4));
                const int sub = __builtin_amdgcn_readfirstlane(*reinterpret_cast<const int*>(ring + kHintStashOffset + 8));
                const uint32_t m8 = (uint32_t)loud & 0xFFu;
                if (m8 >> 7) return;                          // loud up to the last sub-probe: the first level's limit stands
                const int q2 = m8 ? 32 - __builtin_clz(m8) : 0;                           // first sub-probe of the quiet tail
                const uint64_t last2 = (uint64_t)(uint32_t)base2 + (uint64_t)(uint32_t)(q2 + 1) * (uint64_t)(uint32_t)sub - 1u +
                                       (uint32_t)(extra + margin);
                const uint64_t lim2 = (last2 >> 10) + 1u;
                if (lim2 < (uint64_t)(uint32_t)hint_lim) hint_lim = (int)lim2;
                return;
            }
        }
        hint_state |= 2;
        const int step = __builtin_amdgcn_readfirstlane(*reinterpret_cast<const int*>(ring + kHintStashOffset));
        const uint64_t mask = loud;
        if (amp1 == 0 || (mask >> (kProbes - 1))) return;                                  // loud to the very end
        const int q = mask ? 64 - __builtin_clzll(mask) : 0;                               // first probe of the quiet tail
        // 64-bit: (q + 1) * step reaches span + 64 rounds, more than the 2^16-byte headroom of
        // AFSK_MAX_STREAM_LEN leaves below 2^31
        const uint64_t last = (uint64_t)(uint32_t)base + (uint64_t)(uint32_t)(q + 1) * (uint64_t)(uint32_t)step - 1u + (uint32_t)extra;
        const uint64_t lim = (last >> 10) + 1u;
        hint_lim = lim < 0x7fffffffull ? (int)lim : 0x7fffffff;
        if constexpr (REFINE) {
            // the signal ends between probe q - 1 and probe q: eight sub-probes there (one LDS-DMA instruction: lanes 0..7
            // fetch 16 bytes each, every other lane points behind the buffer -- range-checked, no memory request)
            const uint64_t lo64 = (uint64_t)(uint32_t)base + (uint64_t)(uint32_t)q * (uint64_t)(uint32_t)step;
            const int sub = step >> 3;
            if (sub >= 256 && lo64 + (uint64_t)step < 0x7fff0000ull) {
                const int lo = (int)lo64;
                if (lane == 0) {
                    *reinterpret_cast<int*>(ring + kHintStashOffset + 4) = lo;
                    *reinterpret_cast<int*>(ring + kHintStashOffset + 8) = sub;
                }
                const int po = lane < 8 ? lo + (lane + 1) * sub - 16 : 0x7ffffff0;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, AFSK_LDS(ring + kProbeOffset), 16, po, 0, 0, 0);
                eval_need = next;                            // once a chunk requested from here on has landed, so have they
            }
        }
    }
};

// ------------------------------------------------------------------ phase A (fast)
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

constexpr int kFastWaveLdsProduct = kHintStashOffset + 16;      // 18,448 B of LDS per wave

// ---- register re-alignment helpers (phase A sub-windows, phase B pieces) ----
// Re-align 24 dwords (six aligned 16-byte reads) by S bytes into 20 dwords.
template <int S>
__device__ __forceinline__ void realign(const uint32_t (&W)[24], uint32_t (&x)[20]) {
    constexpr int A = S / 4, B = S % 4;
#pragma unroll
    for (int d = 0; d < 20; d++) {
        if constexpr (B == 0) x[d] = W[d + A];
        else x[d] = __builtin_amdgcn_alignbyte(W[d + A + 1], W[d + A], B);
    }
}

// Same for NI aligned dwords -> NO dwords (the 8-byte-aligned pieces of the 2400-baud mapping).
template <int S, int NI, int NO>
__device__ __forceinline__ void realign_n(const uint32_t (&W)[NI], uint32_t (&x)[NO]) {
    constexpr int A = S / 4, B = S % 4;
    static_assert(NO + A + (B ? 1 : 0) <= NI, "not enough input dwords");
#pragma unroll
    for (int d = 0; d < NO; d++) {
        if constexpr (B == 0) x[d] = W[d + A];
        else x[d] = __builtin_amdgcn_alignbyte(W[d + A + 1], W[d + A], B);
    }
}

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
    const uint64_t valid = nv >= 64 ? ~0ull : ((1ull << nv) - 1ull);
    const uint64_t stop = valid & ~((1ull << start) - 1ull) & ~amp_ok;     // start < 64
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
    for (int j0 = d.bytes_done; j0 < jnew; j0 += 64) {
        const int j = j0 + lane;
        bool fix0 = false, fix1 = false;       // soft output: non-zero syndromes (ref:147)
        if (j < jnew) {
            const int g = d.st.term_sym + 14 * j;
            const uint64_t lo = words[(g >> 6) & (kBitWords - 1)];
            const uint64_t hi = words[((g >> 6) + 1) & (kBitWords - 1)];
            const int sh = g & 63;
            uint32_t c = (uint32_t)(lo >> sh);
            if (sh > 50) c |= (uint32_t)(hi << (64 - sh));
            c &= 0x3FFFu;
            const uint32_t byte = (hamming_nibble(c & 127u) << 4) | hamming_nibble(c >> 7);   // ref:393-399
            if (j < out_stride) out_row[j] = (uint8_t)byte;
            fix0 = hamming_syndrome(c & 127u) != 0;
            fix1 = hamming_syndrome(c >> 7) != 0;
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
template <int SPL, class AmpFn>
__device__ __forceinline__ void rxd_round(RxDeferred& d, const uint64_t (&B)[SPL], int32_t K, int k0,
                                          int lane, unsigned long long* words, uint8_t* out_row,
                                          int out_stride, AmpFn&& amp_word) {
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
    if (start >= 0 && start < nv) {                            // squelch stop (ref:372-376)
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
            return compress_bits<LPS>(__ballot(loud_enough(q, (uint32_t)BF, amp_thr)));
        });
    }
}

// The round loop.  ALIGNED = the wave-uniform shift (2*ci) & 15 is zero (always true for
// Transmitter-generated streams, whose clock index is a multiple of the training period):
// five aligned ds_read_b128 feed the arithmetic directly.  Otherwise six reads + v_alignbyte.
template <int BF, int FLAGS, bool ALIGNED, bool HINTED>
__device__ __forceinline__ void fast_rounds(FastRing& fr, int byte0, int32_t K, int32_t NR,
                                            uint32_t amp_thr, RxDeferred& rd,
                                            unsigned long long* words, uint8_t* out_row,
                                            int out_stride, int32_t* margins, int32_t mstride) {
    constexpr int SPR = 2560 / BF;                                // symbols per 5 KiB round
    const int lane = fr.lane;
    const int shift = byte0 & 15;
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
            // all 64 banks.  ALIGNED here means (2*ci) & 7 == 0.
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
                    switch (byte0 & 7) {
                        case 2: realign_n<2, 12, 10>(W, y); break;
                        case 4: realign_n<4, 12, 10>(W, y); break;
                        default: realign_n<6, 12, 10>(W, y); break;
                    }
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
            switch (shift) {
                case 2: realign<2>(W, x); break;
                case 4: realign<4>(W, x); break;
                case 6: realign<6>(W, x); break;
                case 8: realign<8>(W, x); break;
                case 10: realign<10>(W, x); break;
                case 12: realign<12>(W, x); break;
                default: realign<14>(W, x); break;
            }
        }
        // the reads above have returned (their values are in x): refill the 5 chunks this
        // round consumed right away, before the arithmetic
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (HINTED && partial) {
            // (no refill: the round may have to run again on the same ring contents)
        } else if (HINTED && fr.hint_takes_over(5)) {
            fr.template top_up<(FLAGS & 4) ? 0 : 2>(((byte0 + 5120 * (r + 1)) >> 10) + kRingChunks);
        } else {
#pragma unroll
            for (int j = 0; j < 5; j++) fr.template issue<(FLAGS & 4) ? 0 : 2>(fr.next + j);
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

template <int BF, int FLAGS, bool ALIGNED, bool HINTED>
__device__ __forceinline__ void multi_rounds(FastRing& fr, int byte0, int32_t K, int32_t NR,
                                             uint32_t amp_thr, RxDeferred& rd,
                                             unsigned long long* words, uint8_t* out_row,
                                             int out_stride, int32_t* margins, int32_t mstride) {
    using MG = MultiGeom<BF>;
    constexpr int R = MG::R, SPL = MG::SPL, PB = MG::PB, RW = MG::RW, NO = MG::NO, SPR = MG::SPR;
    constexpr int Q = BF / 4, H = BF / 2;
    constexpr uint32_t FULL = 65535u;
    constexpr int NR_READS = PB / RW;                          // reads per piece when aligned
    constexpr int DW = RW / 4;                                 // dwords per read
    const int lane = fr.lane;
    // 32-byte pieces (bit_frames 16): the sixteen lanes of a ds_read_b128 group, 32 bytes apart, pair up on eight bank
    // quads (two-way conflict on every read).  Lanes with bit 3 set read the second 16 bytes of their piece first:
    // the pairs then touch different quads.  Their registers hold the symbol's two halves exchanged, i.e. quarter
    // sums (h2, h3, h0, h1): the mark correlator is symmetric under that exchange, the space correlator flips sign.
#ifndef AFSK_SWAP16
#define AFSK_SWAP16 0
#endif
    constexpr bool SWAP16 = (AFSK_SWAP16) && ALIGNED && BF == 16;
    const bool swap16 = SWAP16 && ((lane >> 3) & 1);
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
            if constexpr (SWAP16) {
                const uint8_t* src_a = src + (swap16 ? 16 : 0);        // read 0 takes the second half in the swapped lanes
                const uint8_t* src_b = src + (swap16 ? 0 : 16);
#pragma unroll
                for (int piece = 0; piece < SPL; piece++) {
                    read_piece(src_a + 64 * PB * piece, piece, 0);
                    read_piece(src_b + 64 * PB * piece, piece, 1);
                }
            } else {
#pragma unroll
            for (int piece = 0; piece < SPL; piece++)
#pragma unroll
                for (int j = 0; j < NR_READS; j++) read_piece(src + 64 * PB * piece + RW * j, piece, j);
            }
            asm volatile("" ::: "memory");                             // (keeps the compiler from merging the two forms into selects)
        } else {
#pragma unroll
        for (int piece = 0; piece < SPL; piece++) {
            const int pb = rb + 64 * PB * piece + PB * lane;
            if constexpr (ALIGNED) {
#pragma unroll
                for (int j = 0; j < NR_READS; j++)
                    read_piece(fr.ring + ((pb + RW * (SWAP16 ? (j ^ (int)swap16) : j)) & (kRingBytes - 1)), piece, j);
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
                switch (byte0 & (RW - 1)) {
                    case 2: realign_n<2, NO + DW, NO>(W, y); break;
                    case 4: realign_n<4, NO + DW, NO>(W, y); break;
                    case 6: realign_n<6, NO + DW, NO>(W, y); break;
                    default:
                        if constexpr (RW == 16) {
                            switch (byte0 & 15) {
                                case 8: realign_n<8, NO + DW, NO>(W, y); break;
                                case 10: realign_n<10, NO + DW, NO>(W, y); break;
                                case 12: realign_n<12, NO + DW, NO>(W, y); break;
                                default: realign_n<14, NO + DW, NO>(W, y); break;
                            }
                        }
                        break;
                }
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
                    if constexpr (SWAP16) {    // exchanged halves: hq = (h2, h3, h0, h1) -> u - 2FQ and dd trade places
                        const uint32_t sp_swapped = 2u * FULL * Q + dd - (hq[0] - hq[3]);
                        space = swap16 ? sp_swapped : space;
                    }
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
        // The decisions of the first EARLY slices are formed while the reads of the later ones are still in flight
        // (ds_reads return in order; the compiler places the partial waits); then, with every value in registers,
        // the consumed chunks are requested again and the rest follows.  EARLY = 0: refill first, as until r5.
#ifndef AFSK_MULTI_EARLY
#define AFSK_MULTI_EARLY 0
#endif
        constexpr int EARLY = (AFSK_MULTI_EARLY) < 0 ? SPL / 2 : ((AFSK_MULTI_EARLY) < SPL ? (AFSK_MULTI_EARLY) : SPL - 1);
        static_for<0, EARLY>(decide);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // values are in x: refill
        if (HINTED && partial) {
            // (no refill: the round may have to run again on the same ring contents)
        } else if (HINTED && fr.hint_takes_over(R)) {
            fr.template top_up<(FLAGS & 4) ? 0 : 2>(((byte0 + 1024 * R * (r + 1)) >> 10) + kRingChunks);
        } else {
#pragma unroll
            for (int j = 0; j < R; j++) fr.template issue<(FLAGS & 4) ? 0 : 2>(fr.next + j);
            fr.next += R;
        }
        static_for<EARLY, SPL>(decide);
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
        if constexpr (SPL == 1) {
            const int nv = (Kr - k0) < 64 ? (Kr - k0) : 64;
            rxd_pass<64>(rd, B[0], nv, k0, lane, words, out_row, out_stride, [&]() { return amp_word(0); });
        } else {
            rxd_round<SPL>(rd, B, Kr, k0, lane, words, out_row, out_stride, amp_word);
        }
        if (rd.st.phase == 2) break;
        if (HINTED && partial) {           // no squelch stop among the symbols that were there: fetch the rest, run the round again
            rd = saved;
            fr.template fetch_through<(FLAGS & 4) ? 0 : 2>(last >> 10);
            r--;
        }
    }
}

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

template <int BF, int FLAGS, bool ALIGNED, bool HINTED>
__device__ __forceinline__ void wm_rounds(FastRing& fr, int byte0, int32_t K, int32_t NR,
                                          uint32_t amp_thr, RxDeferred& rd,
                                          unsigned long long* words, uint8_t* out_row,
                                          int out_stride, int32_t* margins, int32_t mstride) {
    using G = WmGeom<BF>;
    constexpr int LPS = G::LPS, PL = G::PL, PB = G::PB, NO = G::NO, RW = G::RW, SPP = G::SPP, RBYTES = G::RBYTES;
    constexpr int Q = BF / 4, H = BF / 2;
    constexpr uint32_t FULL = 65535u;
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
            switch (byte0 & (RW - 1)) {                           // wave-uniform, even, non-zero
                case 2: realign_n<2, NW, NO>(W, x); break;
                case 4: if constexpr (RW >= 8) realign_n<4, NW, NO>(W, x); break;
                case 6: if constexpr (RW >= 8) realign_n<6, NW, NO>(W, x); break;
                case 8: if constexpr (RW == 16) realign_n<8, NW, NO>(W, x); break;
                case 10: if constexpr (RW == 16) realign_n<10, NW, NO>(W, x); break;
                case 12: if constexpr (RW == 16) realign_n<12, NW, NO>(W, x); break;
                default: if constexpr (RW == 16) realign_n<14, NW, NO>(W, x); break;
            }
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
            uint64_t am = __ballot(loud_enough(q, (uint32_t)BF, amp_thr));
            if constexpr (LPS >= 2) am = compress_bits<LPS>(am);
            return am;
        });
        if (rd.st.phase == 2) break;
        if (HINTED && partial) {           // no squelch stop among the symbols that were there: fetch the rest, run the round again
            rd = saved;
            fr.template fetch_through<(FLAGS & 4) ? 0 : 2>(last >> 10);
            r--; pos -= RBYTES;
        }
    }
}

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
// mirror behind the ring, 2-byte-aligned dword reads (a clock index may be odd).
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

template <int BF, int FLAGS, bool HINTED>
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
        const int sh = byte0 & 2;                                     // wave-uniform: 0, or 2 for an odd clock index
        const uint8_t* src = fr.ring + (((rb + piece_byte) & (kRingBytes - 1)) - sh);
        constexpr int PALIGN = G::PALIGN;
        constexpr int NW = NB + 3;                                    // one dword more for the shifted form
        uint32_t W[NW];
        if (PALIGN >= 8 && (byte0 & (PALIGN - 1)) == 0) {
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
        if (sh == 0) {
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
            return compress_bits_last<LPS>(__ballot(loud_enough(qsum, (uint32_t)BF, amp_thr)));
        });
        if (rd.st.phase == 2) break;
        if (HINTED && partial) {           // no squelch stop among the symbols that were there: fetch the rest, run the round again
            rd = saved;
            fr.template fetch_through<(FLAGS & 4) ? 0 : 2>(last >> 10);
            r--; pos -= RBYTES;
        }
    }
}

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
#pragma unroll
    for (int c = 0; c < PRE; c++) fr.template issue<(FLAGS & 4) ? 0 : 2>(c);
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
    const int byte0 = 2 * ci;                                  // ring byte of symbol 0
    // Tail hint (see kProbes): for large launches, and only on the aligned round loops (a second copy of
    // each, so that streams without the hint run exactly the code they ran before).  bit_frames 4 / 8 (five-
    // and ten-slice rounds, short of scalar registers) take it only in their UNIFORM kernels and only from
    // kHintMinStreamsShort4 / 8 streams on (r4: 12000 baud 0.709 -> 0.750 of peak at 65536 streams, 6000 baud
    // 0.731 -> 0.750; at 4096 streams it costs them 3 %, and in the per-stream kernel, which carries every
    // geometry's scalars, 1 - 2 %).
    constexpr bool HINT = UNIFORM || !(MULTI && MultiGeom<MULTI ? BF : 4>::SPL >= 5);
    constexpr int kAlignMask = GP ? 0 : (WM ? WmGeom<WM ? BF : 60>::RW - 1 : (MULTI ? MultiGeom<MULTI ? BF : 4>::RW - 1 : (BF == 20 ? 7 : 15)));   // GP reads 2-byte-aligned dwords
    const bool aligned = (byte0 & kAlignMask) == 0;               // 2400 baud reads 8-byte pieces
    const bool hinted = HINT && hint && aligned;
    if (hinted) {
        constexpr int kRoundBytes = MULTI ? 1024 * MultiGeom<MULTI ? BF : 4>::R : (WM ? WmGeom<WM ? BF : 60>::RBYTES : (GP ? GpGeom<GP ? BF : 128>::RBYTES : 5120));
        fr.request_probes((uint32_t)len * 2u, byte0, kRoundBytes);
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
#define AFSK_ROUNDS(FN)                                                                                              \
    do {                                                                                                             \
        if (!aligned) FN<BF, FLAGS, false, false>(fr, byte0, K, NR, amp_thr, rd, words, out_row, out_stride, margins, mstride); \
        else if (HINT && BIG && hinted) FN<BF, FLAGS, true, HINT && BIG>(fr, byte0, K, NR, amp_thr, rd, words, out_row, out_stride, margins, mstride); \
        else FN<BF, FLAGS, true, false>(fr, byte0, K, NR, amp_thr, rd, words, out_row, out_stride, margins, mstride);   \
    } while (0)
    if constexpr (GP) {
        if (HINT && BIG && hinted) gp_rounds<BF, FLAGS, HINT && BIG>(fr, byte0, K, NR, amp_thr, rd, words, out_row, out_stride, margins, mstride);
        else gp_rounds<BF, FLAGS, false>(fr, byte0, K, NR, amp_thr, rd, words, out_row, out_stride, margins, mstride);
    } else if constexpr (WM) AFSK_ROUNDS(wm_rounds);
    else if constexpr (MULTI) AFSK_ROUNDS(multi_rounds);
    else AFSK_ROUNDS(fast_rounds);
#undef AFSK_ROUNDS
    rxd_finish<PS>(rd, K, lane, words, out_row, out_stride);
    st = rd.st;
    wait_vmcnt<0>();   // drain DMA still in flight before the LDS region is released
}

}  // namespace afsk
