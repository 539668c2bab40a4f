// afsk_demod.hip -- product instantiation of the batched demodulator kernel.
#include "afsk_demod_impl.h"

namespace afsk {

hipError_t launch_demod(const DemodArgs& a, hipStream_t stream) {
    if (a.n_streams <= 0) return hipSuccess;
    const int blocks = (a.n_streams + kWavesPerBlock - 1) / kWavesPerBlock;
    hipLaunchKernelGGL((demod_kernel_t<0, true>), dim3(blocks), dim3(64 * kWavesPerBlock), 0, stream, a);
    return hipGetLastError();
}

}  // namespace afsk
