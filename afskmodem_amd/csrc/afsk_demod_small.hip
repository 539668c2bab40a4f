// afsk_demod_small.hip -- product instantiation of the demod kernel for launches below
// kHintMinStreams streams (no L2 warming, no tail hint in the code at all).
#include "afsk_demod_impl.h"

namespace afsk {

hipError_t launch_demod_small(const DemodArgs& a, int blocks, hipStream_t stream) {
    hipLaunchKernelGGL((demod_kernel_t<0, kWavesPerBlock, 0, false>), dim3(blocks), dim3(64 * kWavesPerBlock), 0,
                       stream, a);
    return hipGetLastError();
}

}  // namespace afsk
