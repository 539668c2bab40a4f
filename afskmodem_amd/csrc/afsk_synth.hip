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

namespace afsk {

struct pack8 { int16_t v[8]; };
// 16 bytes to a 2-byte-aligned address with ONE global_store_dwordx4 (gfx950 stores are
// alignment-agnostic; a packed struct made hipcc split the store in three).
typedef uint32_t store16 __attribute__((ext_vector_type(4), aligned(2)));

__device__ __forceinline__ void store_pack8(int16_t* dst, const pack8& v) {
    store16 w;
#pragma unroll
    for (int k = 0; k < 4; k++)
        w[k] = (uint32_t)(uint16_t)v.v[2 * k] | ((uint32_t)(uint16_t)v.v[2 * k + 1] << 16);
    *reinterpret_cast<store16*>(dst) = w;
}

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
constexpr int kModIters = 4;                                   // 16-byte stores per thread
constexpr int kModChunk = kModThreads * 8 * kModIters;         // samples per block (8192)

// Branch-free tone kind of symbol S (true = mark).  `win` is the block's payload window in
// LDS: win[k] = payload[first_byte + k].
__device__ __forceinline__ bool symbol_is_mark(uint32_t S, uint32_t n_train_sym, uint32_t first_byte,
                                               const uint8_t* win) {
    const uint32_t t = S - n_train_sym;              // wraps for training symbols (unused then)
    const uint32_t b = t - 4u;                       // coded bit index (wraps before the data)
    const uint32_t cw = b / 7u;                      // nibble index
    const uint32_t pos = b - cw * 7u;
    const uint32_t byte = win[((cw >> 1) - first_byte) & (kModThreads - 1)];
    const uint32_t nib = (cw & 1u) ? (byte & 15u) : (byte >> 4);        // ref:446-450 MSB first
    const bool data_bit = ((hamming_codeword(nib) >> pos) & 1u) != 0;
    const bool train = S < n_train_sym;
    const bool term = t < 4u;
    return train ? ((S & 1u) == 0) : (term ? (t == 0u) : data_bit);
}

constexpr int kQWords = kModChunk / 32 + 16;                  // quarter-symbol bitmap of one block

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

template <bool QUIRK, bool SMALLQ>
__device__ __forceinline__ void tone_block(int16_t* dst0, uint32_t base, uint32_t len, uint32_t phb,
                                           uint32_t q, const uint32_t* qb) {
    const float rcp_q = 1.0f / (float)q;
    const uint32_t mq = (65536u + q - 1u) / q;
#pragma unroll
    for (int it = 0; it < kModIters; it++) {
        const uint32_t local = ((uint32_t)it * kModThreads + threadIdx.x) * 8u;
        const uint32_t p0 = base + local;
        if (p0 >= len) break;
        const store16 w = tone_words<QUIRK, SMALLQ>(phb + local, q, rcp_q, mq, qb);
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

// One block = 8192 consecutive output samples of one stream; one thread = 4 x (8 samples =
// one 16-byte store).  Blocks past the tones store zeros; blocks inside the tones build, once,
// the payload window, a bitmap of the tone kind of every symbol they touch (one ballot per 64
// symbols) and from it the quarter-symbol bitmap tone_words() reads.  Only the block holding
// the tones/silence boundary (and bit_frames that are no multiple of 4) takes the general
// per-frame path below.  Positions fit in 32 bits (stream_len < 2^30).
__global__ __launch_bounds__(kModThreads) void modulate_kernel(ModulateArgs a) {
    __shared__ uint8_t win[kModThreads];
    // tone kind (1 = mark) of every symbol the block touches: at most 8192/4 + 3 symbols
    __shared__ unsigned long long kinds[kModChunk / 4 / 64 + 2];
    __shared__ uint32_t qbits[kQWords];
    const int s = blockIdx.x / a.chunks;
    const int chunk = blockIdx.x - s * a.chunks;
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
    const uint32_t lim = n_tones < n_out ? n_tones : n_out;    // frames at or past this are zero
    int16_t* dst0 = a.samples + a.stream_offset[s];

    if (base >= lim) {                                         // tail silence ref:468 + padding
#pragma unroll
        for (int it = 0; it < kModIters; it++) {
            const uint32_t p0 = base + ((uint32_t)it * kModThreads + threadIdx.x) * 8u;
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

    // payload window of this block: the block spans < 8192 / (14 * bf) + 2 <= 148 bytes
    const uint32_t data0 = n_train_sym + 4u;
    const uint32_t Sb = base / bf;
    const uint32_t first_byte = ((Sb > data0 ? Sb - data0 : 0u) / 7u) >> 1;
    {
        const uint32_t idx = first_byte + threadIdx.x;
        win[threadIdx.x] = idx < plen ? payload[idx] : (uint8_t)0;
    }
    __syncthreads();
    // Kind bitmap: bit r = symbol Sb + r.  Each wave ballots 64 consecutive symbols per pass.
    const uint32_t last = (base + kModChunk - 1u) / bf + 2u;              // exclusive upper bound + slack
    const uint32_t nsym_blk = last - Sb + 1u;
    {
        const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
        for (uint32_t r0 = wave * 64u; r0 < nsym_blk; r0 += kModThreads) {
            const uint32_t S = Sb + r0 + lane;
            const bool mk = S < n_sym && symbol_is_mark(S, n_train_sym, first_byte, win);
            const unsigned long long m = __ballot(mk);
            if (lane == 0) kinds[r0 >> 6] = m;
        }
    }
    __syncthreads();
    const uint32_t phb = base - Sb * bf;

    if ((bf & 3u) == 0u && base + kModChunk + 8u <= lim) {      // block-uniform: all tones
        // quarter bitmap: word w = symbols Sb + 8w .. 8w+7, 4 bits each: space 0b0011, mark 0b0101
        const uint32_t nwords = (nsym_blk + 7u) >> 3;
        for (uint32_t w = threadIdx.x; w <= nwords; w += kModThreads) {
            uint32_t x = w < nwords ? (uint32_t)(kinds[w >> 3] >> ((w & 7u) * 8u)) & 0xFFu : 0u;
            x = (x | (x << 12)) & 0x000F000Fu;                   // bit k -> bit 4k
            x = (x | (x << 6)) & 0x03030303u;
            x = (x | (x << 3)) & 0x11111111u;
            qbits[w] = 0x33333333u ^ (x * 6u);
        }
        __syncthreads();
        const uint32_t q = bf >> 2;
        if (a.wav_quirk) {
            if (q >= 8u) tone_block<true, false>(dst0, base, len, phb, q, qbits);
            else tone_block<true, true>(dst0, base, len, phb, q, qbits);
        } else {
            if (q >= 8u) tone_block<false, false>(dst0, base, len, phb, q, qbits);
            else tone_block<false, true>(dst0, base, len, phb, q, qbits);
        }
        return;
    }

    // General path (the block that contains the end of the tones).  Symbol / phase of this
    // thread's first store, then advanced by 2048 samples per iteration without further
    // divisions: x < bf + 2048 (bf < 2048), so a float estimate + fix-up is exact.
    const float rcp_bf = 1.0f / (float)bf;
    auto divmod_small = [&](uint32_t x, uint32_t& q, uint32_t& r) {
        q = (uint32_t)((float)x * rcp_bf);
        int32_t rr = (int32_t)(x - q * bf);
        if (rr < 0) { q -= 1; rr += (int32_t)bf; }
        else if (rr >= (int32_t)bf) { q += 1; rr -= (int32_t)bf; }
        r = (uint32_t)rr;
    };
    uint32_t dq, ph0;
    divmod_small(phb + 8u * threadIdx.x, dq, ph0);
    uint32_t S0 = Sb + dq;
    uint32_t step_q, step_r;                                   // 2048 = step_q * bf + step_r
    divmod_small(2048u, step_q, step_r);

#pragma unroll 1
    for (int it = 0; it < kModIters; it++) {
        const uint32_t p0 = base + ((uint32_t)it * kModThreads + threadIdx.x) * 8u;
        if (p0 >= len) break;
        const uint32_t rel = S0 - Sb;                            // three consecutive bits of the bitmap
        const unsigned long long w0 = kinds[rel >> 6], w1 = kinds[(rel >> 6) + 1u];
        const uint32_t sh = rel & 63u;
        const uint32_t kb = (uint32_t)(w0 >> sh) | (sh > 61u ? (uint32_t)(w1 << (64u - sh)) : 0u);
        const bool k0 = kb & 1u, k1 = kb & 2u, k2 = kb & 4u;
        // tone of the frame at offset dj from p0 (branch-free)
        auto frame = [&](uint32_t dj) -> int16_t {
            const uint32_t phj = ph0 + dj;                      // < bf + 7 < 3*bf
            const uint32_t w = (uint32_t)(phj >= bf) + (uint32_t)(phj >= 2u * bf);
            const uint32_t ph = phj - w * bf;
            const bool mark = w == 0 ? k0 : (w == 1 ? k1 : k2);
            const uint32_t ph4 = 4u * ph;
            const bool hi_mark = (ph4 < bf) | ((ph4 >= 2u * bf) & (ph4 < 3u * bf));   // ref:80-85
            const bool hi_space = 2u * ph < bf;                                          // ref:68-77
            const int16_t tone = (mark ? hi_mark : hi_space) ? (int16_t)32767 : (int16_t)-32768;
            return (p0 + dj) < lim ? tone : (int16_t)0;
        };
        pack8 v;
        if (a.wav_quirk) {        // block-uniform: out[2i] = out[2i+1] = frames[2i] (ref:239-244)
#pragma unroll
            for (uint32_t j = 0; j < 8; j += 2) v.v[j] = v.v[j + 1] = frame(j);
        } else {
#pragma unroll
            for (uint32_t j = 0; j < 8; j++) v.v[j] = frame(j);
        }
        int16_t* dst = dst0 + p0;
        if (p0 + 8u <= len) {
            store_pack8(dst, v);
        } else {
#pragma unroll
            for (uint32_t j = 0; j < 8u; j++)
                if (p0 + j < len) dst[j] = v.v[j];
        }
        // next store of this thread is 2048 samples further
        S0 += step_q;
        ph0 += step_r;
        if (ph0 >= bf) { ph0 -= bf; S0 += 1u; }
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

hipError_t launch_modulate(ModulateArgs a, int32_t max_len, hipStream_t stream) {
    if (a.n_streams <= 0 || max_len <= 0) return hipSuccess;
    const int per_block = kModChunk;
    a.chunks = (max_len + per_block - 1) / per_block;
    const int64_t blocks = (int64_t)a.chunks * a.n_streams;
    if (blocks > 0x7fffffffll) return hipErrorInvalidValue;
    hipLaunchKernelGGL(modulate_kernel, dim3((uint32_t)blocks), dim3(kModThreads), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_noise(NoiseArgs a, int32_t max_len, hipStream_t stream) {
    if (a.n_streams <= 0 || max_len <= 0) return hipSuccess;
    const int per_block = 256 * 8;
    a.chunks = (max_len + per_block - 1) / per_block;
    const int64_t blocks = (int64_t)a.chunks * a.n_streams;
    if (blocks > 0x7fffffffll) return hipErrorInvalidValue;
    hipLaunchKernelGGL(noise_kernel, dim3((uint32_t)blocks), dim3(256), 0, stream, a);
    return hipGetLastError();
}

}  // namespace afsk
