// afsk_demod_ring.h -- part of the single-pass demodulator (afsk_demod_fast.h includes the parts in order; see its header
// comment for the overall design): the 16 KiB LDS-DMA ring of one wave (FastRing): chunk requests, the wait-count invariant, L2 warming,
// the two-level tail hint and partial rounds, the bias-free squelch amplitude, register re-alignment helpers.
#pragma once

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
// from 16384 on (0.69 -> 0.71; 32768: 0.69 -> 0.76) -- profiles/archive/r4_exp4_hint_short.txt
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

// compile-time loop: f(std::integral_constant<int, I>) for I in [I, N)
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

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

    // N consecutive chunks from chunk c on (r5).  The instruction's 12-bit immediate offset moves BOTH ends of an
    // LDS-DMA transfer (global address and LDS address = M0 + offset + 16 * lane), so while the N ring slots do not
    // wrap -- three rounds of four with five chunks per round -- ONE M0 and ONE scalar offset serve four requests
    // (offset:0 / 1024 / 2048 / 3072): 1 + 1 scalar instructions per four chunks instead of three or four per
    // chunk (slot mask, M0, the hazard nop, the next offset).
    template <int AUX, int N>
    __device__ __forceinline__ void issue_run(int c) {
        const int slot = c & (kRingChunks - 1);
        if (slot + N <= kRingChunks) {
            static_for<0, N>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, AFSK_LDS(ring + slot * 1024 + (j / 4) * 4096), 16, lane * 16,
                                                         c * 1024 + (j / 4) * 4096, (j % 4) * 1024, AUX);
            });
        } else {
#pragma unroll
            for (int j = 0; j < N; j++) issue<AUX>(c + j);
        }
    }

    // ---- re-basing the ring on the clock index (r6) ----
    // The ring is requested at stream byte 0 before the clock index is known, and symbol 0 then sits at ring byte
    // 2 * ci: any even number.  The round loops read lane pieces with 16- (8-, 4-) byte LDS reads, which must be that
    // aligned to run at speed (gfx950 executes misaligned ds_read_b128 several times slower), so until r5 a stream
    // with (2 * ci) & 15 != 0 -- 7 of 8 captures that do not start on the first sample of the training sequence, e.g.
    // every burst the live gate cuts at a 2048-sample block boundary (ref:299-319) -- ran a second form of every round
    // loop: one LDS read more per piece and a v_alignbyte per dword behind a seven-way switch, in every round, and no
    // tail hint (-9 % at 1200 baud: profiles/r6_exp1_lead_baseline.txt).  Now the stream is RE-BASED once, right after
    // phase A, by S = (2 * ci) & 12 bytes -- the whole dwords of the misalignment:
    //   * the ring contents from the clock index's chunk on move down by S bytes in place (aligned 16-byte reads of a
    //     lane's own and its neighbour's 16 bytes, a register renaming, one 16-byte write; chunk by chunk, front to back:
    //     a chunk's reads reach 16 bytes into the next one, which is not written before the next iteration);
    //   * the last S bytes of slot 15 -- stream bytes [16384, 16384 + S), never requested -- arrive by a four-lane
    //     dword LDS-DMA into the last 16 bytes of the ring;
    //   * the buffer descriptor moves up by S bytes (base + S, num_records - S): chunk c is now stream bytes
    //     [S + 1024 c, S + 1024 (c + 1)), range checks and tail clipping as before.
    // From here on symbol 0 sits at ring byte 2 * ci - S = a multiple of 16, or -- an odd clock index -- 2 bytes past one,
    // and every stream runs one of TWO forms of its round loop (ODD = the shift by one sample in registers), both with
    // the tail hint, whose probes are requested after this.
    // Why not the last 2 bytes as well: moving the descriptor by an odd number of samples makes every LDS-DMA request
    // 2-byte-aligned in memory, and those run at well under half speed -- 65536 x 1200 baud with every clock index odd:
    // 1064 us against 855 us for S = 4 / 8 / 12 and 848 us for S = 0 (profiles/r6_exp3_rebase_by_shift.txt); the
    // LDS-DMA cannot place data at half a dword in LDS either.
    // Wait-count invariant: the four-lane request completes chunk 15, so it is the one request after which the auxiliary
    // count restarts (warm_ops = 0: the warming requests in front of it are older than it; counting them no longer only
    // makes a wait for a chunk below 15 wait for up to two landed requests more).  A wait for chunk 15 then allows
    // exactly the requests issued behind this one to be outstanding, and the fixed-schedule immediates hold as before
    // (a round that reads chunk 15 has at least 15 - R younger chunk requests behind it).
    template <int A>                                           // A = S / 4 dwords
    __device__ __forceinline__ void shift_down(int c0) {
        static_assert(A >= 1 && A <= 3, "one to three dwords");
        for (int c = c0; c < kRingChunks; c++) {
            uint8_t* p = ring + 1024 * c + 16 * lane;
            const u32x4 lo4 = *reinterpret_cast<const u32x4*>(p);
            const u32x4 hi4 = *reinterpret_cast<const u32x4*>(p + 16);   // (lane 63 of chunk 15: the mirror, whatever it holds)
            const uint32_t W[8] = {lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
            u32x4 o;
#pragma unroll
            for (int d = 0; d < 4; d++) o[d] = W[d + A];
            wave_lds_sync();                                   // every lane has read before any lane writes
            *reinterpret_cast<u32x4*>(p) = o;
            wave_lds_sync();
        }
    }
    // xs / len: the stream as the caller gave it; byte0 = 2 * ci.  Returns byte0 - S, the re-based ring byte of symbol 0.
    template <int AUX>
    __device__ __forceinline__ int rebase(const int16_t* xs, int32_t len, int byte0) {
        const int S = byte0 & 12;
        if (S == 0) return byte0;
        // phase A waited for chunks 0..7 only: everything that moves must have landed (and nothing may land on top of
        // moved bytes later) -- all 16 chunk requests; the warming requests behind them write to their dummy area
        wait_exact<0>(kRingChunks - 1);
        const int c0 = byte0 >> 10;                            // chunks below the clock index are free: not moved
        switch (S) {
            case 4: shift_down<1>(c0); break;
            case 8: shift_down<2>(c0); break;
            default: shift_down<3>(c0); break;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the moved bytes are in place before the DMA below can land
        rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<const uint8_t*>(xs) + S), 0, len * 2 - S, 0x00020000);
        if (lane < 4)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, AFSK_LDS(ring + kRingBytes - 16), 4, lane * 4, kRingBytes - 16, 0, AUX);
        warm_ops = 0;
        return byte0 - S;
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
            issue_run<AUX, CMIN>(next);
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

}  // namespace afsk
