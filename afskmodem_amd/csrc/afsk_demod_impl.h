// afsk_demod_impl.h -- device code of the batched AFSK demodulator for MI355X (gfx950 /
// CDNA4).  Included by the kernel translation units (afsk_demod_small.hip / afsk_demod_big.hip:
// the mixed-baud kernel; afsk_demod_uniform.hip: one kernel per bit_frames) and by tools/kbench.hip.
//
// One wavefront (64 lanes) owns one stream and runs the whole receiver hot path of
// lavajuno/afskmodem for it (reference afskmodem.py, "ref:" below): clock recovery
// ref:322-339, per-symbol mark/space decisions ref:342-351 with the limiter ref:287-296,
// training-terminator scan ref:361-366/386-390, squelch stop ref:372-378 with the amplitude
// of ref:94-98, Hamming(7,4) decode ref:145-163 and MSB-first byte pack ref:393-399.
//
// afsk_demod_fast.h holds the single-pass LDS-DMA ring path, THE product path for every valid
// bit_frames: compile-time geometries for the reference's documented 300 - 12000 baud range
// (4 ... 64, 60, 80, 96, 100, 120, 160; the list is AFSK_FAST_BF_LIST below) and a run-time
// geometry for every other valid value.  Every sample is fetched from HBM exactly once.
//
// Two kernel families share it:
//   * demod_kernel_t           bit_frames PER STREAM (afsk_demod_batch): one body with all geometries
//                              behind a wave-uniform switch -- a mixed-baud batch is one launch;
//   * demod_uniform_kernel_t   ONE bit_frames for the whole launch (afsk_demod_batch_uniform: a
//                              Receiver has exactly one baud rate, ref:275-284), a compile-time
//                              constant of the kernel: no per-stream load, no switch, ~1/20 of the code;
//     demod_uniform_rt_kernel_t  the same for the run-time geometry.
//
// (The round-1 two-pass design lives in tools/afsk_twopass.h as the baseline of tools/kbench.)
//
// No MFMA: this is an HBM-bound streaming reduction (2 B read per sample).
// No workgroup barrier: the 4 waves of a block are independent streams.
#include <hip/hip_runtime.h>

#include <stdint.h>

#include "afsk_kernels.h"

// gfx950 (MI355X) only: LDS-DMA (buffer_load ... lds, dwordx4), DPP row_bcast, SDWA with an SGPR destination, and
// inline asm whose hand-placed wait states follow gfx940 / gfx950's VALU-SGPR hazard rules (afsk_demod_phasec.h,
// afsk_demod_rounds_multi.h).  build.sh refuses any other AFSK_ARCH; this stops a direct hipcc call as well.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "the afsk demod kernels are gfx950 code (see build.sh)"
#endif

namespace afsk {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef short i16x2 __attribute__((ext_vector_type(2)));

#define AFSK_LDS(p) ((__attribute__((address_space(3))) void*)(p))

constexpr int kSync = 4096;          // ref:323,327
constexpr int kWavesPerBlock = 4;
constexpr uint32_t kBias = 0x80008000u;  // int16 -> order-preserving uint16, packed pair

// Lanes of one wave exchange data through LDS (one lane reads what another lane wrote).
// The hardware keeps a wave's LDS operations in order, but the COMPILER reasons per thread:
// without a fence it may delete a store that the same lane never reads back, or hoist a load
// above another lane's store.  This emits no instruction; it only pins compiler ordering.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Inclusive wave scan (64 lanes) with DPP: 4 row_shr steps + row_bcast:15 + row_bcast:31.
__device__ __forceinline__ int32_t wave_incl_scan_dpp(int32_t v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);   // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);   // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);   // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);   // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1,3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2,3
    return v;
}

// Minimum over the 64 lanes, returned wave-uniform.  DPP only (row_shr 1/2/4/8 inside the rows of 16,
// then row_bcast:15 / :31 across rows; lanes without a source keep the identity): six VALU steps and
// one v_readlane -- __shfl_xor would be six ds_bpermute round trips through the LDS pipe.
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
    const int id = (int)0xFFFFFFFFu;
    auto mn = [](uint32_t a, uint32_t b) { return a < b ? a : b; };
    v = mn(v, (uint32_t)__builtin_amdgcn_update_dpp(id, (int)v, 0x111, 0xf, 0xf, false));   // row_shr:1
    v = mn(v, (uint32_t)__builtin_amdgcn_update_dpp(id, (int)v, 0x112, 0xf, 0xf, false));   // row_shr:2
    v = mn(v, (uint32_t)__builtin_amdgcn_update_dpp(id, (int)v, 0x114, 0xf, 0xf, false));   // row_shr:4
    v = mn(v, (uint32_t)__builtin_amdgcn_update_dpp(id, (int)v, 0x118, 0xf, 0xf, false));   // row_shr:8
    v = mn(v, (uint32_t)__builtin_amdgcn_update_dpp(id, (int)v, 0x142, 0xa, 0xf, false));   // row_bcast:15 -> rows 1, 3
    v = mn(v, (uint32_t)__builtin_amdgcn_update_dpp(id, (int)v, 0x143, 0xc, 0xf, false));   // row_bcast:31 -> rows 2, 3
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// floor(total / n) exactly (float estimate + one fix-up step) while the quotient stays below
// 2^16, i.e. total <= 65535 * n -- true for every SAD total here (n <= 4094 samples of at most
// 65535 each), although the totals themselves reach ~2^28.
__device__ __forceinline__ uint32_t div_exact(uint32_t total, uint32_t n, float rcp_n) {
    uint32_t q = (uint32_t)((float)total * rcp_n);
    int32_t r = (int32_t)(total - q * n);
    if (r < 0) q -= 1;
    else if (r >= (int32_t)n) q += 1;
    return q;
}

// Limiter ref:287-296 on a packed pair, result biased by 0x8000 per half:
// x > 512 -> 0xFFFF (32767), x < -512 -> 0x0000 (-32768), else 0x8000 (0).
__device__ __forceinline__ uint32_t limit_pair_biased(uint32_t x) {
    const i16x2 xv = __builtin_bit_cast(i16x2, x);
    const i16x2 k512 = {512, 512};
    const i16x2 km513 = {-513, -513};
    i16x2 ps = __builtin_elementwise_sub_sat(k512, xv) >> 15;    // 0xFFFF iff x > 512
    i16x2 nn = __builtin_elementwise_sub_sat(km513, xv) >> 15;   // 0xFFFF iff x >= -512
    return (__builtin_bit_cast(uint32_t, nn) & kBias) | __builtin_bit_cast(uint32_t, ps);
}

// Hamming(7,4) syndrome (error position, 0 = clean) ref:146-147; bit t of cw = received bit t.
__device__ __forceinline__ uint32_t hamming_syndrome(uint32_t cw) {
    uint32_t s0 = __popc(cw & 0x55u) & 1u;   // parity row 1010101 (ref:126)
    uint32_t s1 = __popc(cw & 0x66u) & 1u;   // parity row 0110011 (ref:127)
    uint32_t s2 = __popc(cw & 0x78u) & 1u;   // parity row 0001111 (ref:128)
    return s2 * 4u + s1 * 2u + s0;           // ref:147
}

// Hamming(7,4) syndrome decode ref:145-151; bit t of cw = received bit t.
__device__ __forceinline__ uint32_t hamming_nibble(uint32_t cw) {
    uint32_t pos = hamming_syndrome(cw);
    if (pos) cw ^= 1u << (pos - 1u);         // ref:149-150
    return (((cw >> 2) & 1u) << 3) | (((cw >> 4) & 1u) << 2) | (((cw >> 5) & 1u) << 1) |
           ((cw >> 6) & 1u);                 // ref:151
}

// ---------------------------------------------------------------- phase C state
// Wave-uniform receiver state machine fed 64 symbol decisions at a time.
struct RxState {
    int phase;        // 0 = skipping training, 1 = data, 2 = stopped
    uint32_t hist;    // last three training decisions: bit0 = oldest
    int32_t nbits;    // coded bits taken (ref:380)
    int32_t nbytes;   // decoded bytes produced
    int32_t term_sym; // symbol index after the terminator (ref:368), -1 = none
    uint32_t pend;    // coded bits not yet forming a byte (< 14 of them)
    int npend;
    int32_t corrected; // codewords decoded so far whose syndrome was non-zero (soft output)
};

// Training part (ref:361-366, 386-390).  bits: bit j = decision of symbol k0+j (j < nv).
// Returns the index of the first DATA symbol inside this pass (0..nv; nv = none of them),
// or -1 while the terminator has not been seen / after the squelch stop.
__device__ __forceinline__ int rx_training(RxState& st, uint64_t bits, int nv, int k0) {
    if (st.phase == 1) return 0;
    if (st.phase != 0) return -1;
    const uint64_t valid = nv >= 64 ? ~0ull : ((1ull << nv) - 1ull);
    bits &= valid;
    // window (b[k-3], b[k-2], b[k-1], b[k]) == (1,0,0,0)
    const uint64_t h = st.hist;
    const uint64_t b3 = (bits << 3) | (h & 7ull);
    const uint64_t b2 = (bits << 2) | ((h >> 1) & 3ull);
    const uint64_t b1 = (bits << 1) | ((h >> 2) & 1ull);
    const uint64_t hit = b3 & ~b2 & ~b1 & ~bits & valid;
    if (hit) {
        const int j = __builtin_ctzll(hit);
        st.term_sym = k0 + j + 1;
        st.phase = 1;
        return j + 1;
    }
    st.hist = nv >= 3 ? (uint32_t)((bits >> (nv - 3)) & 7ull)
                      : (uint32_t)(((h | (bits << 3)) >> nv) & 7ull);
    return -1;
}

// ------------------------------------------------------------------- phase B
// Template dwords (biased) for the two samples at symbol phases ph, ph+1.
__host__ __device__ constexpr uint32_t mark_half(int ph, int q) {
    return ((ph / q) & 1) ? 0x0000u : 0xFFFFu;   // hi on quarters 0 and 2, ref:80-85
}
__host__ __device__ constexpr uint32_t space_half(int ph, int h) {
    return ph < h ? 0xFFFFu : 0x0000u;           // hi on the first half, ref:68-77
}

__device__ __forceinline__ void wait_vmcnt_dyn(int n) {          // n is wave-uniform
    switch (n) {
#define AFSK_W(k) case k: wait_vmcnt<k>(); break;
        AFSK_W(0) AFSK_W(1) AFSK_W(2) AFSK_W(3) AFSK_W(4) AFSK_W(5) AFSK_W(6) AFSK_W(7)
        AFSK_W(8) AFSK_W(9) AFSK_W(10) AFSK_W(11) AFSK_W(12) AFSK_W(13) AFSK_W(14) AFSK_W(15)
        AFSK_W(16) AFSK_W(17) AFSK_W(18) AFSK_W(19) AFSK_W(20) AFSK_W(21) AFSK_W(22) AFSK_W(23)
        AFSK_W(24) AFSK_W(25) AFSK_W(26) AFSK_W(27) AFSK_W(28) AFSK_W(29) AFSK_W(30)
#undef AFSK_W
        default: if (n < 0) wait_vmcnt<0>(); else wait_vmcnt<31>(); break;
    }
}

}  // namespace afsk
#include "afsk_demod_fast.h"
namespace afsk {

// FLAGS are diagnostic only (tools/kbench.hip); the product instantiates FLAGS = 0.
constexpr int kFlagSkipSync = 1;    // force clock index 0, no phase A (results wrong unless ci == 0)
constexpr int kFlagSkipValu = 2;    // phase B streams the ring but skips the per-sample VALU work
constexpr int kFlagNoNt = 4;        // default cache policy instead of non-temporal (nt) ring DMA loads
// (FLAGS & 64: per-wave s_memrealtime stamps into DemodArgs::debug_stamps)

// every bit_frames with a compile-time geometry (everything else: the run-time geometry)
#define AFSK_FAST_BF_LIST(X) X(4) X(8) X(12) X(16) X(20) X(24) X(32) X(40) X(48) X(60) X(64) X(80) X(96) X(100) X(120) X(160) X(240) X(320) X(480)

// bit_frames whose compile-time geometry is built from the general pieces (afsk_demod_fast.h):
// the other values a Receiver can be built for -- 48000 / baud a divisor of 48000 and a multiple of 4 --
// i.e. 375, 250, 240, 160, 125, 120, 96, 80, 75, 60, 50, 48, 40, 32, 30, 25 and 24 baud.  Until r3 a
// mixed-baud launch ran them on the run-time geometry (0.15 - 0.5 of the HBM peak); since r4 the per-stream
// switch of demod_kernel_t covers both lists (0.61 - 0.73 on such mixes, +30 s of build time, 97 instead of 63
// SGPR spills; config #3 unchanged at 0.81: profiles/EXPERIMENTS.md).
#define AFSK_GP_BF_LIST(X) X(128) X(192) X(200) X(300) X(384) X(400) X(500) X(600) X(640) X(800) X(960) X(1000) X(1200) X(1500) X(1600) X(1920) X(2000)

// how many uniform kernels the build must produce (build.sh scrapes the two lists above and checks its
// count against this line, so a reformatted macro cannot silently drop kernels from the library)
#define AFSK_X(B) +1
constexpr int kUniformBfCount = 0 AFSK_FAST_BF_LIST(AFSK_X) AFSK_GP_BF_LIST(AFSK_X);
#undef AFSK_X
static_assert(kUniformBfCount == 36, "every bit_frames a Receiver can be built for has a compile-time geometry");

__host__ __device__ constexpr bool bit_frames_valid(int bf) {       // ref:68-85, 327: templates exist, sync window fits
    return bf >= 4 && (bf & 3) == 0 && 2 * bf < kSync;
}
__host__ __device__ constexpr bool has_fast_geometry(int bf) {
#define AFSK_X(B) if (bf == B) return true;
    AFSK_FAST_BF_LIST(AFSK_X)
#undef AFSK_X
    return false;
}
__host__ __device__ constexpr bool has_uniform_geometry(int bf) {     // compile-time geometry in a uniform kernel
#define AFSK_X(B) if (bf == B) return true;
    AFSK_GP_BF_LIST(AFSK_X)
#undef AFSK_X
    return has_fast_geometry(bf);
}

// streams the decoder refuses before touching a sample: status 3 (invalid bit_frames; the host
// validates first, this only keeps the kernel memory-safe), 4 (stream_len negative or above
// AFSK_MAX_STREAM_LEN: device-side arrays the host never saw) or 1 (shorter than the sync window, ref:323-325)
__device__ __forceinline__ void store_refusal(const DemodArgs& a, int s, int lane, int32_t status) {
    if (lane == 0) {
        a.out_nbytes[s] = 0; a.out_nbits[s] = 0; a.out_clock_idx[s] = -1;
        a.out_term_frame[s] = -1; a.out_status[s] = status;
    }
}

__device__ __forceinline__ void store_result(const DemodArgs& a, int s, int lane, const RxState& st, int ci,
                                             int32_t n_sym, int bf) {
    if (lane == 0) {
        const int32_t term_sym = st.term_sym >= 0 ? st.term_sym : n_sym;   // ref:362-368
        a.out_nbytes[s] = st.nbytes;
        a.out_nbits[s] = st.nbits;
        a.out_clock_idx[s] = ci;
        a.out_term_frame[s] = ci + term_sym * bf;
        a.out_status[s] = st.nbits == 0 ? 2 : 0;   // ref:422-424
        if (a.out_corrected) a.out_corrected[s] = st.corrected;
    }
}

// One stream of a MIXED-baud launch, start to finish, by one wave: bit_frames is a per-stream value.
template <int FLAGS, bool BIG = true>
__device__ __forceinline__ void process_stream(const DemodArgs& a, int s, int64_t off, int32_t len,
                                               int bf, uint8_t* lds, int lane) {
    const int16_t* xs = a.samples + off;
    uint8_t* out_row = a.out_bytes + (int64_t)s * a.out_stride;
    if (!bit_frames_valid(bf)) { store_refusal(a, s, lane, 3); return; }
    if ((uint32_t)len > (uint32_t)kMaxStreamLen) { store_refusal(a, s, lane, 4); return; }   // negative or beyond 32-bit byte offsets
    if (len < kSync) { store_refusal(a, s, lane, 1); return; }
    RxState st;
    int32_t n_sym = 0;
    int32_t* margins = a.out_margins ? a.out_margins + (int64_t)s * a.margin_stride : nullptr;
    int ci = 0;
    unsigned long long* stamps = (FLAGS & 64) ? a.debug_stamps + 4 * s : nullptr;
    const bool warm = a.n_streams >= kWarmMinStreams;          // wave-uniform (afsk_demod_fast.h)
    const bool hint = a.n_streams >= (a.stream_index ? kHintMinStreamsGrouped : kHintMinStreams);
    switch (bf) {
#define AFSK_X(B) case B: demod_stream_fast<B, FLAGS, BIG>(xs, len, a.amp_end, lds, lane, st, out_row, a.out_stride, ci, n_sym, stamps, margins, a.margin_stride, warm, hint); break;
        AFSK_FAST_BF_LIST(AFSK_X)
        AFSK_GP_BF_LIST(AFSK_X)      // r4: every rate a Receiver can be built for has its compile-time geometry here too
#undef AFSK_X
        default:            // every other valid bit_frames: the run-time geometry on the same ring
            demod_stream_rt<FLAGS, BIG>(xs, len, bf, a.amp_end, lds, lane, st, out_row, a.out_stride, ci, n_sym,
                                        margins, a.margin_stride, warm, hint);
            break;
    }
    store_result(a, s, lane, st, ci, n_sym, bf);
}

// One wave per stream, one block per 4 streams; the hardware dispatcher balances blocks over
// the CUs (a persistent grid with static striding measured 5 % slower at 65536 streams and
// no faster at 4096, so it is not used).
// WPB = waves (= streams) per block; LDS_PER_WAVE >= kFastWaveLdsProduct sets how many blocks fit a
// CU's 160 KiB LDS, i.e. the number of resident waves per CU.
// BIG = the kernel for launches of kHintMinStreams or more (compiled WITH the large-launch measures:
// L2 warming, tail hint); the other instantiation does not even contain their tests -- the extra live
// scalars cost config #2 1.5 % when both lived in one piece of code.  launch_demod picks the kernel.
template <int FLAGS, int WPB = kWavesPerBlock, int LDS_PER_WAVE = 0, bool BIG = false>
__global__ __launch_bounds__(64 * WPB) void demod_kernel_t(DemodArgs a) {
    constexpr int kLdsPerWave = LDS_PER_WAVE > 0 ? LDS_PER_WAVE : kFastWaveLdsProduct;
    static_assert(kLdsPerWave >= kFastWaveLdsProduct, "LDS per wave too small");
    __shared__ __attribute__((aligned(16))) uint8_t lds_all[WPB * kLdsPerWave];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int w = xcd_block((int)blockIdx.x, (int)gridDim.x) * WPB + wave;
    if (w >= a.n_streams) return;
    // grouped dispatch, fused form: the launch walks a rate-sorted LIST of streams (neighbouring waves run the
    // same geometry's code); bit_frames[] and every other per-stream array stay indexed by the stream number
    const int s = a.stream_index ? __builtin_amdgcn_readfirstlane(a.stream_index[w]) : w;
    if constexpr (FLAGS & 64) {   // diagnostic build: wall-clock stamps (100 MHz s_memrealtime)
        if (lane == 0) a.debug_stamps[4 * s + 0] = __builtin_amdgcn_s_memrealtime();
    }
    process_stream<FLAGS, BIG>(a, s, a.stream_offset[s], a.stream_len[s], a.bit_frames[s],
                               lds_all + wave * kLdsPerWave, lane);
    if constexpr (FLAGS & 64) {
        if (lane == 0) a.debug_stamps[4 * s + 3] = __builtin_amdgcn_s_memrealtime();
    }
}

// launches of this many streams or more run the large-launch form of a uniform kernel (tail hint; L2 warming
// from kWarmMinStreams); bit_frames 4 / 8 later than the others (kHintMinStreamsShort4 / 8)
__host__ __device__ constexpr int uniform_big_from(int bf) {
    return bf == 4 ? kHintMinStreamsShort4 : (bf == 8 ? kHintMinStreamsShort8 : kHintMinStreamsUniform);
}

// ---- one bit_frames for the whole launch (afsk_demod_batch_uniform) ------------------------------
// BF > 0: the compile-time geometry of that value; BF == 0: the run-time geometry with
// a.uniform_bit_frames (host-validated: valid and without a compile-time geometry).
// one stream of a uniform launch, start to finish, by one wave
template <int BF, int FLAGS, bool BIG>
__device__ __forceinline__ void process_uniform_stream(const DemodArgs& a, int s, uint8_t* lds, int lane) {
    const int32_t len = a.stream_len[s];
    // a DEVICE-side length the host never saw: negative, or beyond what 32-bit byte offsets address
    if ((uint32_t)len > (uint32_t)kMaxStreamLen) { store_refusal(a, s, lane, 4); return; }
    if (len < kSync) { store_refusal(a, s, lane, 1); return; }   // ref:323-325
    const int16_t* xs = a.samples + a.stream_offset[s];
    uint8_t* out_row = a.out_bytes + (int64_t)s * a.out_stride;
    int32_t* margins = a.out_margins ? a.out_margins + (int64_t)s * a.margin_stride : nullptr;
    const bool warm = a.n_streams >= kWarmMinStreams;
    const bool hint = a.n_streams >= uniform_big_from(BF);
    RxState st;
    int32_t n_sym = 0;
    int ci = 0;
    unsigned long long* stamps = (FLAGS & 64) ? a.debug_stamps + 4 * s : nullptr;   // diagnostic build only
    if constexpr (FLAGS & 64) {
        if (lane == 0) stamps[0] = __builtin_amdgcn_s_memrealtime();
    }
    if constexpr (BF > 0) {
        demod_stream_fast<BF, FLAGS, BIG, true>(xs, len, a.amp_end, lds, lane, st, out_row, a.out_stride, ci, n_sym, stamps,
                                                margins, a.margin_stride, warm, hint);
        store_result(a, s, lane, st, ci, n_sym, BF);
    } else {
        const int bf = a.uniform_bit_frames;
        demod_stream_rt<FLAGS, BIG>(xs, len, bf, a.amp_end, lds, lane, st, out_row, a.out_stride, ci, n_sym, margins,
                                    a.margin_stride, warm, hint);
        store_result(a, s, lane, st, ci, n_sym, bf);
    }
    if constexpr (FLAGS & 64) {
        if (lane == 0) stamps[3] = __builtin_amdgcn_s_memrealtime();
    }
}

template <int BF, int FLAGS = 0, bool BIG = false, int WPB = kWavesPerBlock>
__global__ __launch_bounds__(64 * WPB) void demod_uniform_kernel_t(DemodArgs a) {
    static_assert(BF == 0 || has_uniform_geometry(BF), "no compile-time geometry for this bit_frames");
    __shared__ __attribute__((aligned(16))) uint8_t lds_all[WPB * kFastWaveLdsProduct];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int w = xcd_block((int)blockIdx.x, (int)gridDim.x) * WPB + wave;
    if (w >= a.n_streams) return;
    // a ragged one-rate batch walks a length-sorted list of its streams (r6: GroupPlan::bucket, afsk_capi.hip)
    const int s = a.stream_index ? __builtin_amdgcn_readfirstlane(a.stream_index[w]) : w;
    process_uniform_stream<BF, FLAGS, BIG>(a, s, lds_all + wave * kFastWaveLdsProduct, lane);
}

}  // namespace afsk
