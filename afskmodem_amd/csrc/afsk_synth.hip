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

struct __attribute__((packed, aligned(2))) pack8 { int16_t v[8]; };

// ref:115-123: codeword p1 p2 d1 p3 d2 d3 d4 for nibble d1 d2 d3 d4 (d1 = MSB)
__device__ __forceinline__ uint32_t hamming_encode_bit(uint32_t nib, int pos) {
    const uint32_t d1 = (nib >> 3) & 1u, d2 = (nib >> 2) & 1u, d3 = (nib >> 1) & 1u, d4 = nib & 1u;
    switch (pos) {
        case 0: return d1 ^ d2 ^ d4;   // row 1101
        case 1: return d1 ^ d3 ^ d4;   // row 1011
        case 2: return d1;             // row 1000
        case 3: return d2 ^ d3 ^ d4;   // row 0111
        case 4: return d2;             // row 0100
        case 5: return d3;             // row 0010
        default: return d4;            // row 0001
    }
}

// Symbol layout of the ideal frame list (ref:452-469), in symbols of bf frames:
//   [0, 2*ts)            training cycles: mark, space, mark, space ...        ref:457-458
//   [2*ts, 2*ts + 4)     terminator: mark, space, space, space               ref:460-462
//   [2*ts + 4, n_sym)    one symbol per Hamming-coded payload bit            ref:463-467
// followed by 4800 zero frames (ref:468) and zero padding.
struct SymbolMap {
    uint32_t n_train_sym, n_sym;
    const uint8_t* payload;

    // true = mark tone, false = space tone, for symbol index S < n_sym
    __device__ __forceinline__ bool is_mark(uint32_t S, uint32_t byte_lo, uint32_t byte_hi,
                                            uint32_t first_byte) const {
        if (S < n_train_sym) return (S & 1u) == 0;
        const uint32_t t = S - n_train_sym;
        if (t < 4) return t == 0;
        const uint32_t b = t - 4;                  // coded bit index
        const uint32_t cw = b / 7u;                // nibble index
        const uint32_t pos = b - cw * 7u;
        const uint32_t byte = ((cw >> 1) == first_byte) ? byte_lo : byte_hi;
        const uint32_t nib = (cw & 1u) ? (byte & 15u) : (byte >> 4);   // ref:446-450 MSB first
        return hamming_encode_bit(nib, (int)pos) != 0;
    }
};

// One thread = 8 consecutive output samples = one 16-byte store.  All positions fit in
// 32 bits (stream_len < 2^30).  One integer division locates the first frame's symbol; the
// thread touches at most 3 symbols (bf >= 4), whose tone kinds are resolved up front.
__global__ __launch_bounds__(256) void modulate_kernel(ModulateArgs a) {
    const int s = blockIdx.x / a.chunks;
    const int chunk = blockIdx.x - s * a.chunks;
    const uint32_t len = (uint32_t)a.stream_len[s];
    const uint32_t p0 = ((uint32_t)chunk * blockDim.x + threadIdx.x) * 8u;
    if (p0 >= len) return;
    const uint32_t bf = (uint32_t)a.bit_frames[s];
    const uint32_t plen = (uint32_t)a.payload_len[s];
    SymbolMap m;
    m.n_train_sym = 2u * (uint32_t)a.ts_cycles[s];
    m.n_sym = m.n_train_sym + 4u + 14u * plen;
    m.payload = a.payload + (int64_t)s * a.payload_stride;
    const uint64_t n_tones64 = (uint64_t)m.n_sym * bf;
    const uint64_t n_frames64 = n_tones64 + 4800u;
    // wav quirk ref:239-244: out[2i] = out[2i+1] = frames[2i] for 2i < n_frames - 1
    const uint64_t n_out64 = a.wav_quirk ? (n_frames64 & ~1ull) : n_frames64;
    const uint32_t n_tones = n_tones64 > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)n_tones64;
    const uint32_t n_out = n_out64 > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)n_out64;
    int16_t* dst = a.samples + a.stream_offset[s] + p0;

    const uint32_t S0 = p0 / bf;
    uint32_t ph = p0 - S0 * bf;
    // payload bytes the (at most three) symbols S0..S0+2 can need
    uint32_t first_byte = 0, byte_lo = 0, byte_hi = 0;
    if (S0 + 2 >= m.n_train_sym + 4u && plen > 0) {
        const uint32_t b0 = S0 > m.n_train_sym + 4u ? S0 - (m.n_train_sym + 4u) : 0u;
        first_byte = (b0 / 7u) >> 1;
        if (first_byte < plen) byte_lo = m.payload[first_byte];
        if (first_byte + 1 < plen) byte_hi = m.payload[first_byte + 1];
    }
    bool kind[3];
#pragma unroll
    for (int k = 0; k < 3; k++)
        kind[k] = (S0 + k < m.n_sym) ? m.is_mark(S0 + k, byte_lo, byte_hi, first_byte) : false;

    const uint32_t step = a.wav_quirk ? 2u : 1u;
    pack8 v;
    uint32_t wraps = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const uint32_t p = p0 + j;
        const bool fresh = a.wav_quirk ? ((j & 1) == 0) : true;   // odd outputs repeat the even frame
        if (fresh) {
            int16_t val = 0;
            if (p < n_tones && p < n_out) {
                const bool mark = wraps == 0 ? kind[0] : (wraps == 1 ? kind[1] : kind[2]);
                const uint32_t ph4 = 4u * ph;
                const bool hi_mark = (ph4 < bf) || (ph4 >= 2u * bf && ph4 < 3u * bf);   // ref:80-85
                const bool hi_space = 2u * ph < bf;                                      // ref:68-77
                val = (mark ? hi_mark : hi_space) ? (int16_t)32767 : (int16_t)-32768;
            }
            v.v[j] = val;
            ph += step;
            if (ph >= bf) { ph -= bf; wraps++; }
            if (ph >= bf) { ph -= bf; wraps++; }   // step 2 with bf == ... keeps ph < bf (bf >= 4)
        } else {
            v.v[j] = (p < n_out) ? v.v[j - 1] : (int16_t)0;
        }
    }
    if (p0 + 8u <= len) {
        *reinterpret_cast<pack8*>(dst) = v;
    } else {
        for (uint32_t j = 0; j < 8u && p0 + j < len; j++) dst[j] = v.v[j];
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
    const int per_block = 256 * 8;
    a.chunks = (max_len + per_block - 1) / per_block;
    const int64_t blocks = (int64_t)a.chunks * a.n_streams;
    if (blocks > 0x7fffffffll) return hipErrorInvalidValue;
    hipLaunchKernelGGL(modulate_kernel, dim3((uint32_t)blocks), dim3(256), 0, stream, a);
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
