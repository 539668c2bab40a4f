// afsk_demod_big.hip -- product instantiation of the demod kernel for launches of kHintMinStreams
// streams or more (L2 warming behind the ring start, tail hint).
#include "afsk_demod_impl.h"

namespace afsk {

hipError_t launch_demod_big(const DemodArgs& a, int blocks, hipStream_t stream) {
    hipLaunchKernelGGL((demod_kernel_t<0, kWavesPerBlock, 0, true>), dim3(blocks), dim3(64 * kWavesPerBlock), 0,
                       stream, a);
    return hipGetLastError();
}

}  // namespace afsk
