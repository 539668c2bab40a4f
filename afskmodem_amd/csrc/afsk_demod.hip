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

}  // namespace afsk
