// afsk_synth.hip -- on-device input synthesis for the batched demodulator.
//
//   modulate_kernel : Transmitter.__getFrames (ref afskmodem.py:452-469) with
//                     ECC.encode (ref:166-175, generator rows ref:115-123) and the
//                     .wav writer's decimate-by-2 + duplicate (ref:239-244).
//   noise_kernel    : build-owned deterministic integer noise (no reference
//                     counterpart), same arithmetic as oracle afsk_o_add_noise.
//
// Both are write-bandwidth bound (2 B per sample); one thread produces 8
// consecutive samples and stores them with one 16-byte store.
#include "afsk_kernels.h"

#include <type_traits>

namespace afsk {

// 16 bytes to a 2-byte-aligned address with ONE global_store_dwordx4 (gfx950 stores are
// alignment-agnostic; a packed struct made hipcc split the store in three).
typedef uint32_t store16 __attribute__((ext_vector_type(4), aligned(2)));

// Symbol layout of the ideal frame list (ref:452-469), in symbols of bf frames:
//   [0, 2*ts)            training cycles: mark, space, mark, space ...        ref:457-458
//   [2*ts, 2*ts + 4)     terminator: mark, space, space, space               ref:460-462
//   [2*ts + 4, n_sym)    one symbol per Hamming-coded payload bit            ref:463-467
// followed by 4800 zero frames (ref:468) and zero padding.

// Hamming(7,4) codeword of a nibble as a 7-bit integer, bit k = k-th transmitted bit
// (p1 p2 d1 p3 d2 d3 d4; generator rows ref:115-123).
__device__ __forceinline__ uint32_t hamming_codeword(uint32_t nib) {
    const uint32_t d1 = (nib >> 3) & 1u, d2 = (nib >> 2) & 1u, d3 = (nib >> 1) & 1u, d4 = nib & 1u;
    return (d1 ^ d2 ^ d4) | ((d1 ^ d3 ^ d4) << 1) | (d1 << 2) | ((d2 ^ d3 ^ d4) << 3) | (d2 << 4) |
           (d3 << 5) | (d4 << 6);
}

constexpr int kModThreads = 256;
constexpr int kModIters = 8;                                   // 16-byte stores per thread (product)
constexpr int mod_chunk(int iters, int threads = kModThreads) { return threads * 8 * iters; }   // samples per block
constexpr int kWinBytes = 512;                                 // payload window of one block (power of two)

// Branch-free tone kind of symbol S (true = mark).  `win` is the block's payload window in
// LDS: win[k] = payload[first_byte + k].
__device__ __forceinline__ bool symbol_is_mark(uint32_t S, uint32_t n_train_sym, uint32_t first_byte,
                                               const uint8_t* win) {
    const uint32_t t = S - n_train_sym;              // wraps for training symbols (unused then)
    const uint32_t b = t - 4u;                       // coded bit index (wraps before the data)
    const uint32_t cw = b / 7u;                      // nibble index
    const uint32_t pos = b - cw * 7u;
    const uint32_t byte = win[((cw >> 1) - first_byte) & (kWinBytes - 1)];
    const uint32_t nib = (cw & 1u) ? (byte & 15u) : (byte >> 4);        // ref:446-450 MSB first
    const bool data_bit = ((hamming_codeword(nib) >> pos) & 1u) != 0;
    const bool train = S < n_train_sym;
    const bool term = t < 4u;
    return train ? ((S & 1u) == 0) : (term ? (t == 0u) : data_bit);
}

constexpr int q_words(int chunk) { return chunk / 32 + 16; }   // quarter-symbol bitmap words of a block

constexpr uint32_t kHi2 = 0x7FFF7FFFu, kLo2 = 0x80008000u;    // two samples at +32767 / -32768

// 8 samples of a block that lies entirely inside the tones.  A tone is constant over a quarter
// symbol (q = bf/4 frames: space = hi,hi,lo,lo ref:68-77, mark = hi,lo,hi,lo ref:80-85), so the
// block keeps ONE bit per quarter symbol in LDS (qb, bit r = quarter 4*Sb + r is high) and a
// store needs the quarter of its first frame (one exact float division: x0 < 2^14) plus
//   q >= 8 : at most one quarter boundary inside the 8 frames -> compare against its position
//   q <  8 : the quarter offset of every frame by a 16-bit reciprocal multiply (exact below 14)
template <bool QUIRK, bool SMALLQ>
__device__ __forceinline__ store16 tone_words(uint32_t x0, uint32_t q, float rcp_q, uint32_t mq,
                                              const uint32_t* qb) {
    const uint32_t Q0 = (uint32_t)(((float)x0 + 0.5f) * rcp_q);
    const uint32_t r0 = x0 - Q0 * q;
    const uint32_t bits = __builtin_amdgcn_alignbit(qb[(Q0 >> 5) + 1u], qb[Q0 >> 5], Q0 & 31u);
    store16 w;
    if constexpr (!SMALLQ) {
        const uint32_t c = q - r0;                               // frames j < c are in quarter Q0
        if constexpr (QUIRK) {                                   // out[2i] = out[2i+1] = frames[2i]
            const uint32_t d0 = (bits & 1u) ? kHi2 : kLo2, d1 = (bits & 2u) ? kHi2 : kLo2;
#pragma unroll
            for (uint32_t d = 0; d < 4; d++) w[d] = 2u * d < c ? d0 : d1;
        } else {
            const uint32_t l0 = (bits & 1u) ? 0x7FFFu : 0x8000u, l1 = (bits & 2u) ? 0x7FFFu : 0x8000u;
            const uint32_t h0 = l0 << 16, h1 = l1 << 16;
#pragma unroll
            for (uint32_t d = 0; d < 4; d++) w[d] = (2u * d < c ? l0 : l1) | (2u * d + 1u < c ? h0 : h1);
        }
    } else {
        auto hi = [&](uint32_t j) -> bool {
            const uint32_t dq = ((r0 + j) * mq) >> 16;           // (r0 + j) / q, r0 + j <= 13
            return (bits >> dq) & 1u;
        };
#pragma unroll
        for (uint32_t d = 0; d < 4; d++) {
            if constexpr (QUIRK) w[d] = hi(2u * d) ? kHi2 : kLo2;
            else w[d] = (hi(2u * d) ? 0x7FFFu : 0x8000u) | (hi(2u * d + 1u) ? 0x7FFF0000u : 0x80000000u);
        }
    }
    return w;
}

// TAIL = the block contains the end of the tones: frames at or past `lim` are zero (lim is even
// when bit_frames % 4 == 0, so whole dwords switch).
template <int ITERS, int THREADS, bool QUIRK, bool SMALLQ, bool TAIL>
__device__ __forceinline__ void tone_block(int16_t* dst0, uint32_t base, uint32_t len, uint32_t phb,
                                           uint32_t q, const uint32_t* qb, uint32_t lim) {
    const float rcp_q = 1.0f / (float)q;
    const uint32_t mq = (65536u + q - 1u) / q;
#pragma unroll
    for (int it = 0; it < ITERS; it++) {
        const uint32_t local = ((uint32_t)it * THREADS + threadIdx.x) * 8u;
        const uint32_t p0 = base + local;
        if (p0 >= len) break;
        store16 w = tone_words<QUIRK, SMALLQ>(phb + local, q, rcp_q, mq, qb);
        if constexpr (TAIL) {
#pragma unroll
            for (uint32_t d = 0; d < 4; d++) w[d] = p0 + 2u * d < lim ? w[d] : 0u;
        }
        int16_t* dst = dst0 + p0;
        if (p0 + 8u <= len) {
            *reinterpret_cast<store16*>(dst) = w;
        } else {
#pragma unroll
            for (uint32_t j = 0; j < 8u; j++)
                if (p0 + j < len) dst[j] = (int16_t)(w[j >> 1] >> (16u * (j & 1u)));
        }
    }
}

// One block = ITERS * 2048 consecutive output samples of one stream; one thread = ITERS x
// (8 samples = one 16-byte store).  Blocks past the tones store zeros; the others build, once,
// the payload window (skipped inside the training sequence), a bitmap of the tone kind of every
// symbol they touch (one ballot per 64 symbols) and from it the quarter-symbol bitmap
// tone_words() reads.  Positions fit in 32 bits (stream_len < 2^30).
template <int ITERS, int THREADS = kModThreads>
__global__ __launch_bounds__(THREADS) void modulate_kernel_t(ModulateArgs a) {
    constexpr int kModChunk = mod_chunk(ITERS, THREADS);
    static_assert(kModChunk / 56 + 4 <= kWinBytes, "payload window too small for the block");
    __shared__ uint8_t win[kWinBytes];
    // tone kind (1 = mark) of every symbol the block touches: at most chunk/4 + 3 symbols
    __shared__ unsigned long long kinds[kModChunk / 4 / 64 + 2];
    __shared__ uint32_t qbits[q_words(kModChunk)];
    // (xcd_block: an XCD writes 8 consecutive 16 KiB chunks instead of every eighth one -- +1.2 %, profiles/r5_exp33_modulator_xcd.txt)
    const int bid = xcd_block((int)blockIdx.x, (int)gridDim.x);
    const int s = bid / a.chunks;
    const int chunk = bid - s * a.chunks;
    // a device-side length outside [0, max_stream_len] (the caller's own bound, which sized the grid) is
    // refused: nothing is written for that stream
    if ((uint32_t)a.stream_len[s] > (uint32_t)a.max_len) return;
    const uint32_t len = (uint32_t)a.stream_len[s];
    const uint32_t base = (uint32_t)chunk * kModChunk;
    if (base >= len) return;                                   // block-uniform
    const uint32_t bf = (uint32_t)a.bit_frames[s];
    const uint32_t plen = (uint32_t)a.payload_len[s];
    const uint32_t n_train_sym = 2u * (uint32_t)a.ts_cycles[s];
    const uint32_t n_sym = n_train_sym + 4u + 14u * plen;
    const uint8_t* payload = a.payload + (int64_t)s * a.payload_stride;
    const uint64_t n_tones64 = (uint64_t)n_sym * bf;
    const uint64_t n_frames64 = n_tones64 + 4800u;
    // wav quirk ref:239-244: out[2i] = out[2i+1] = frames[2i] for 2i < n_frames - 1
    const uint64_t n_out64 = a.wav_quirk ? (n_frames64 & ~1ull) : n_frames64;
    const uint32_t n_tones = n_tones64 > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)n_tones64;
    const uint32_t n_out = n_out64 > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)n_out64;
    // frames at or past lim are zero; a bit_frames outside the kernels' domain (not a positive
    // multiple of 4, include/afsk_amd.h) yields an all-zero stream
    const bool bf_ok = bf >= 4u && bf <= (1u << 20) && (bf & 3u) == 0u;   // negative wraps high
    const uint32_t lim = bf_ok ? (n_tones < n_out ? n_tones : n_out) : 0u;
    int16_t* dst0 = a.samples + a.stream_offset[s];

    if (base >= lim) {                                         // tail silence ref:468 + padding
#pragma unroll
        for (int it = 0; it < ITERS; it++) {
            const uint32_t p0 = base + ((uint32_t)it * THREADS + threadIdx.x) * 8u;
            if (p0 >= len) break;
            int16_t* dst = dst0 + p0;
            if (p0 + 8u <= len) {
                *reinterpret_cast<store16*>(dst) = store16{0u, 0u, 0u, 0u};
            } else {
                for (uint32_t j = 0; j < 8u; j++)
                    if (p0 + j < len) dst[j] = 0;
            }
        }
        return;
    }

    // payload window of this block: the block spans < chunk / (14 * bf) + 2 bytes
    const uint32_t data0 = n_train_sym + 4u;
    const uint32_t Sb = base / bf;
    const uint32_t first_byte = ((Sb > data0 ? Sb - data0 : 0u) / 7u) >> 1;
    const uint32_t last = (base + kModChunk - 1u) / bf + 2u;              // exclusive upper bound + slack
    const uint32_t nsym_blk = last - Sb + 1u;
    if (Sb + nsym_blk > data0) {                                           // block-uniform: data symbols
        for (uint32_t k = threadIdx.x; k < (uint32_t)kWinBytes; k += THREADS) {
            const uint32_t idx = first_byte + k;
            win[k] = idx < plen ? payload[idx] : (uint8_t)0;
        }
        __syncthreads();
    }
    // Kind bitmap: bit r = symbol Sb + r.  Each wave ballots 64 consecutive symbols per pass.
    {
        const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
        for (uint32_t r0 = wave * 64u; r0 < nsym_blk; r0 += THREADS) {
            const uint32_t S = Sb + r0 + lane;
            const bool mk = S < n_sym && symbol_is_mark(S, n_train_sym, first_byte, win);
            const unsigned long long m = __ballot(mk);
            if (lane == 0) kinds[r0 >> 6] = m;
        }
    }
    __syncthreads();
    const uint32_t phb = base - Sb * bf;

    {
        // quarter bitmap: word w = symbols Sb + 8w .. 8w+7, 4 bits each: space 0b0011, mark 0b0101
        const uint32_t nwords = (nsym_blk + 7u) >> 3;
        for (uint32_t w = threadIdx.x; w <= nwords; w += THREADS) {
            uint32_t x = w < nwords ? (uint32_t)(kinds[w >> 3] >> ((w & 7u) * 8u)) & 0xFFu : 0u;
            x = (x | (x << 12)) & 0x000F000Fu;                   // bit k -> bit 4k
            x = (x | (x << 6)) & 0x03030303u;
            x = (x | (x << 3)) & 0x11111111u;
            qbits[w] = 0x33333333u ^ (x * 6u);
        }
        __syncthreads();
        const uint32_t q = bf >> 2;
        const bool tail = base + kModChunk + 8u > lim;          // the tones end inside this block
        auto run = [&](auto quirk, auto smallq) {
            if (tail) tone_block<ITERS, THREADS, decltype(quirk)::value, decltype(smallq)::value, true>(dst0, base, len, phb, q, qbits, lim);
            else tone_block<ITERS, THREADS, decltype(quirk)::value, decltype(smallq)::value, false>(dst0, base, len, phb, q, qbits, lim);
        };
        if (a.wav_quirk) {
            if (q >= 8u) run(std::true_type{}, std::false_type{});
            else run(std::true_type{}, std::true_type{});
        } else {
            if (q >= 8u) run(std::false_type{}, std::false_type{});
            else run(std::false_type{}, std::true_type{});
        }
    }
}

__device__ __forceinline__ uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

__global__ __launch_bounds__(256) void noise_kernel(NoiseArgs a) {
    const int s = blockIdx.x / a.chunks;
    const int chunk = blockIdx.x - s * a.chunks;
    const int32_t len = a.stream_len[s];
    if ((uint32_t)len > (uint32_t)a.max_len) return;           // refused like modulate_kernel_t: stream untouched
    const int64_t p0 = ((int64_t)chunk * blockDim.x + threadIdx.x) * 8;
    if (p0 >= len) return;
    const int64_t scale = a.scale_q24[s];
    const uint32_t key = hash32(a.seed ^ hash32(a.stream_idx_base + (uint32_t)s + 0x9e3779b9U));
    int16_t* dst = a.samples + a.stream_offset[s] + p0;
    for (int j = 0; j < 8 && p0 + j < len; j++) {
        const uint32_t t = (uint32_t)(p0 + j);
        int32_t sum = 0;
#pragma unroll
        for (uint32_t k = 0; k < 8; k++) {
            const uint32_t hsh = hash32(key ^ (t * 8u + k));
            sum += (int32_t)(hsh & 0xffffu) + (int32_t)(hsh >> 16);
        }
        const int64_t centred = (int64_t)sum - 524280;
        const int64_t noise = (centred * scale + (1 << 23)) >> 24;
        int64_t v = (int64_t)dst[j] + noise;
        v = v > 32767 ? 32767 : (v < -32768 ? -32768 : v);
        dst[j] = (int16_t)v;
    }
}

template <int ITERS, int THREADS = kModThreads>
hipError_t launch_modulate_t(ModulateArgs a, int32_t max_len, hipStream_t stream) {
    if (a.n_streams <= 0 || max_len <= 0) return hipSuccess;
    const int per_block = mod_chunk(ITERS, THREADS);
    a.chunks = (max_len + per_block - 1) / per_block;
    a.max_len = max_len;
    const int64_t blocks = (int64_t)a.chunks * a.n_streams;
    if (blocks > 0x7fffffffll) return hipErrorInvalidValue;
    hipLaunchKernelGGL((modulate_kernel_t<ITERS, THREADS>), dim3((uint32_t)blocks), dim3(THREADS), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_modulate(ModulateArgs a, int32_t max_len, hipStream_t stream) {
    return launch_modulate_t<kModIters>(a, max_len, stream);
}

hipError_t launch_noise(NoiseArgs a, int32_t max_len, hipStream_t stream) {
    if (a.n_streams <= 0 || max_len <= 0) return hipSuccess;
    const int per_block = 256 * 8;
    a.chunks = (max_len + per_block - 1) / per_block;
    a.max_len = max_len;
    const int64_t blocks = (int64_t)a.chunks * a.n_streams;
    if (blocks > 0x7fffffffll) return hipErrorInvalidValue;
    hipLaunchKernelGGL(noise_kernel, dim3((uint32_t)blocks), dim3(256), 0, stream, a);
    return hipGetLastError();
}

}  // namespace afsk
