// afsk_demod.hip -- launch of the batched demodulator.
//   launch_demod          bit_frames per stream: two product kernels (small / large launches), each
//                         instantiated in its own translation unit (afsk_demod_small.hip / _big.hip);
//   launch_demod_uniform  one bit_frames for the launch: one kernel per compile-time geometry (and
//                         one for the run-time geometry), small and large form each, instantiated by
//                         afsk_demod_uniform.hip compiled once per value (build.sh: -DAFSK_UNIFORM_BF=N).
#include "afsk_demod_impl.h"

namespace afsk {

hipError_t launch_demod_small(const DemodArgs& a, int blocks, hipStream_t stream);
hipError_t launch_demod_big(const DemodArgs& a, int blocks, hipStream_t stream);
#define AFSK_X(B) hipError_t launch_demod_uniform_##B(const DemodArgs& a, int blocks, bool big, hipStream_t stream);
AFSK_FAST_BF_LIST(AFSK_X)
AFSK_GP_BF_LIST(AFSK_X)
AFSK_X(0)
#undef AFSK_X

hipError_t launch_demod(const DemodArgs& a, hipStream_t stream) {
    if (a.n_streams <= 0) return hipSuccess;
    const int blocks = (a.n_streams + kWavesPerBlock - 1) / kWavesPerBlock;
    // launches of kHintMinStreams or more run the kernel with the large-launch measures (L2 warming
    // from kWarmMinStreams, tail hint); smaller ones a kernel compiled without them
    // (a rate-sorted stream list -- the grouped dispatch -- takes them from kHintMinStreamsGrouped on)
    const int big_from = a.stream_index ? kHintMinStreamsGrouped : kHintMinStreams;
    return a.n_streams >= big_from ? launch_demod_big(a, blocks, stream) : launch_demod_small(a, blocks, stream);
}

hipError_t launch_demod_uniform(const DemodArgs& a, hipStream_t stream) {
    if (a.n_streams <= 0) return hipSuccess;
    if (!bit_frames_valid(a.uniform_bit_frames)) return hipErrorInvalidValue;   // the C-ABI entry checks first
    const int blocks = (a.n_streams + kWavesPerBlock - 1) / kWavesPerBlock;
    const bool big = a.n_streams >= uniform_big_from(a.uniform_bit_frames);
    switch (a.uniform_bit_frames) {
#define AFSK_X(B) case B: return launch_demod_uniform_##B(a, blocks, big, stream);
        AFSK_FAST_BF_LIST(AFSK_X)
        AFSK_GP_BF_LIST(AFSK_X)
#undef AFSK_X
        default: return launch_demod_uniform_0(a, blocks, big, stream);       // run-time geometry
    }
}

// ---- results in list order -> stream order (grouped dispatch, staged form) ----------------------------------------
// One workgroup per 32 consecutive streams of the OUTPUT: their status words are one 128-byte line per array and their
// rows are neighbours, so everything this kernel writes leaves the L2 as whole, merged lines -- which is the point: a
// rate-sorted walk writing at the stream numbers itself touches every line once per rate, minutes of kernel time apart
// in cache terms (profiles/r5_exp35_rate_order_and_partial_lines.txt).
constexpr int kPermuteStreams = 32;

__global__ __launch_bounds__(256) void permute_results_kernel(PermuteArgs a) {
    __shared__ int32_t pos[kPermuteStreams];
    __shared__ int32_t nbs[kPermuteStreams];
    const int t = (int)threadIdx.x;
    const int s0 = (int)blockIdx.x * kPermuteStreams;
    if (t < kPermuteStreams) {
        const int s = s0 + t;
        int32_t w = 0, nb = 0;
        if (s < a.n) {
            w = a.inv[s];
            const int32_t nbytes = a.st_nbytes[w];
            a.out_nbytes[s] = nbytes;
            a.out_nbits[s] = a.st_nbits[w];
            a.out_clock_idx[s] = a.st_clock_idx[w];
            a.out_term_frame[s] = a.st_term_frame[w];
            a.out_status[s] = a.st_status[w];
            if (a.out_corrected) a.out_corrected[s] = a.st_corrected[w];
            nb = nbytes < 0 ? 0 : (nbytes < a.stride ? nbytes : a.stride);
        }
        pos[t] = w;
        nbs[t] = nb;
    }
    __syncthreads();
    const int dw_per_row = a.stride >> 2;
    const int total = kPermuteStreams * dw_per_row;
    for (int idx = t; idx < total; idx += 256) {
        const int r = idx / dw_per_row;
        const int j = 4 * (idx - r * dw_per_row);
        const int s = s0 + r;
        if (s >= a.n) break;
        const int nb = nbs[r];
        if (j >= nb) continue;
        const uint8_t* src = a.st_bytes + (int64_t)pos[r] * a.stride + j;
        uint8_t* dst = a.out_bytes + (int64_t)s * a.stride + j;
        if (j + 4 <= nb) {
            *reinterpret_cast<uint32_t*>(dst) = *reinterpret_cast<const uint32_t*>(src);
        } else {
            for (int k = 0; k < nb - j; k++) dst[k] = src[k];
        }
    }
}

hipError_t launch_permute_results(const PermuteArgs& a, hipStream_t stream) {
    if (a.n <= 0) return hipSuccess;
    hipLaunchKernelGGL(permute_results_kernel, dim3((uint32_t)((a.n + kPermuteStreams - 1) / kPermuteStreams)), dim3(256), 0, stream, a);
    return hipGetLastError();
}

}  // namespace afsk
