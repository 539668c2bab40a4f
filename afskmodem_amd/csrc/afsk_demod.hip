// afsk_demod.hip -- launch of the batched demodulator: two product kernels, each instantiated in its
// own translation unit (afsk_demod_small.hip / afsk_demod_big.hip) so that they compile in parallel.
#include "afsk_demod_impl.h"

namespace afsk {

hipError_t launch_demod_small(const DemodArgs& a, int blocks, hipStream_t stream);
hipError_t launch_demod_big(const DemodArgs& a, int blocks, hipStream_t stream);

hipError_t launch_demod(const DemodArgs& a, hipStream_t stream) {
    if (a.n_streams <= 0) return hipSuccess;
    const int blocks = (a.n_streams + kWavesPerBlock - 1) / kWavesPerBlock;
    // launches of kHintMinStreams or more run the kernel with the large-launch measures (L2 warming
    // from kWarmMinStreams, tail hint); smaller ones a kernel compiled without them
    return a.n_streams >= kHintMinStreams ? launch_demod_big(a, blocks, stream) : launch_demod_small(a, blocks, stream);
}

}  // namespace afsk
