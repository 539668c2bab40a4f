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

__device__ __forceinline__ int16_t tone_sample(bool mark, int ph, int q, int h) {
    const bool hi = mark ? (((ph / q) & 1) == 0) : (ph < h);   // ref:68-85
    return hi ? (int16_t)32767 : (int16_t)-32768;
}

__device__ __forceinline__ int16_t frame_value(int64_t f, int bf, int64_t n_train,
                                               int64_t n_total_tones, const uint8_t* payload) {
    // f indexes the ideal frame list of ref:452-469 (before the wav quirk)
    const int q = bf >> 2, h = bf >> 1;
    if (f < n_train) {                       // ref:457-458 training cycles: mark, space
        const int ph = (int)(f % (2 * bf));
        return ph < bf ? tone_sample(true, ph, q, h) : tone_sample(false, ph - bf, q, h);
    }
    if (f >= n_total_tones) return 0;        // ref:468 tail silence (and zero padding)
    const int64_t g = f - n_train;
    const int64_t sym = g / bf;
    const int ph = (int)(g - sym * bf);
    if (sym < 4) return tone_sample(sym == 0, ph, q, h);   // ref:460-462 terminator
    const int64_t b = sym - 4;               // coded bit index, ref:463-467
    const int64_t cw = b / 7;
    const int pos = (int)(b - cw * 7);
    const uint8_t byte = payload[cw >> 1];
    const uint32_t nib = (cw & 1) ? (byte & 15u) : (byte >> 4);   // ref:446-450 MSB first
    return tone_sample(hamming_encode_bit(nib, pos) != 0, ph, q, h);
}

__global__ __launch_bounds__(256) void modulate_kernel(ModulateArgs a) {
    const int s = blockIdx.x / a.chunks;
    const int chunk = blockIdx.x - s * a.chunks;
    const int32_t len = a.stream_len[s];
    const int64_t p0 = ((int64_t)chunk * blockDim.x + threadIdx.x) * 8;
    if (p0 >= len) return;
    const int bf = a.bit_frames[s];
    const int64_t n_train = (int64_t)a.ts_cycles[s] * 2 * bf;
    const int64_t n_tones = n_train + (int64_t)(4 + 14 * (int64_t)a.payload_len[s]) * bf;
    const int64_t n_frames = n_tones + 4800;
    // wav quirk ref:239-244: out[2i] = out[2i+1] = frames[2i] for 2i < n_frames - 1
    const int64_t n_out = a.wav_quirk ? (n_frames & ~1ll) : n_frames;
    const uint8_t* payload = a.payload + (int64_t)s * a.payload_stride;
    int16_t* dst = a.samples + a.stream_offset[s] + p0;
    pack8 v;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int64_t p = p0 + j;
        int16_t val = 0;
        if (p < n_out) val = frame_value(a.wav_quirk ? (p & ~1ll) : p, bf, n_train, n_tones, payload);
        v.v[j] = val;
    }
    if (p0 + 8 <= len) {
        *reinterpret_cast<pack8*>(dst) = v;
    } else {
        for (int j = 0; j < 8 && p0 + j < len; j++) dst[j] = v.v[j];
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
