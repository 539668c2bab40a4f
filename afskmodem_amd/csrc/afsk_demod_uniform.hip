// afsk_demod_uniform.hip -- the uniform-baud kernels of ONE bit_frames value (small- and large-launch
// form).  build.sh compiles this file once per value of AFSK_FAST_BF_LIST with -DAFSK_UNIFORM_BF=<value>
// and once with -DAFSK_UNIFORM_BF=0 (the run-time geometry), all in parallel.
#include "afsk_demod_impl.h"

#ifndef AFSK_UNIFORM_BF
#error "compile with -DAFSK_UNIFORM_BF=<bit_frames> (0 = run-time geometry)"
#endif

#define AFSK_CAT_(a, b) a##b
#define AFSK_CAT(a, b) AFSK_CAT_(a, b)

namespace afsk {

hipError_t AFSK_CAT(launch_demod_uniform_, AFSK_UNIFORM_BF)(const DemodArgs& a, int blocks, bool big, hipStream_t stream) {
    if (big)
        hipLaunchKernelGGL((demod_uniform_kernel_t<AFSK_UNIFORM_BF, 0, true>), dim3(blocks), dim3(64 * kWavesPerBlock), 0, stream, a);
    else
        hipLaunchKernelGGL((demod_uniform_kernel_t<AFSK_UNIFORM_BF, 0, false>), dim3(blocks), dim3(64 * kWavesPerBlock), 0, stream, a);
    return hipGetLastError();
}

}  // namespace afsk
