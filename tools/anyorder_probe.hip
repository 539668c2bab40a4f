// anyorder_probe.hip -- how do several small kernels of one "step" overlap on gfx950?
//   (a) back to back on one stream; (b) the same with hipExtAnyOrderLaunch on all but the first;
//   (c) forked onto side streams with events and joined back (what GroupPlan::launch does).
// Each kernel: 256 blocks x 256 threads, 72 KiB of LDS per block (2 blocks per CU, like the demod kernels),
// every wave busy-waits `us` microseconds.   hipcc -O3 --offload-arch=gfx950 -o anyorder_probe anyorder_probe.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ __launch_bounds__(256) void spin_kernel(int ticks, int* sink) {
    __shared__ int lds[72 * 1024 / 4];
    lds[threadIdx.x] = threadIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();   // 100 MHz
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)ticks) __builtin_amdgcn_s_sleep(8);
    if (lds[(threadIdx.x * 7) & 255] == -1) *sink = 1;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

int main(int argc, char** argv) {
    const int us = argc > 1 ? std::atoi(argv[1]) : 30, kernels = argc > 2 ? std::atoi(argv[2]) : 4;
    const int blocks = argc > 3 ? std::atoi(argv[3]) : 256, steps = 200;
    int* sink; CK(hipMalloc(&sink, 4));
    hipStream_t main_s; CK(hipStreamCreateWithFlags(&main_s, hipStreamNonBlocking));
    std::vector<hipStream_t> side(kernels); std::vector<hipEvent_t> join(kernels);
    hipEvent_t fork, t0, t1; CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming)); CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
    for (int k = 0; k < kernels; k++) { CK(hipStreamCreateWithFlags(&side[k], hipStreamNonBlocking)); CK(hipEventCreateWithFlags(&join[k], hipEventDisableTiming)); }
    auto timed = [&](const char* name, auto&& step) {
        for (int i = 0; i < 20; i++) step();
        CK(hipStreamSynchronize(main_s));
        CK(hipEventRecord(t0, main_s));
        for (int i = 0; i < steps; i++) step();
        CK(hipEventRecord(t1, main_s));
        CK(hipStreamSynchronize(main_s));
        float ms = 0; CK(hipEventElapsedTime(&ms, t0, t1));
        std::printf("%-34s %7.1f us per step (%d kernels x %d blocks x %d us)\n", name, ms * 1e3 / steps, kernels, blocks, us);
    };
    timed("one stream, in order", [&] {
        for (int k = 0; k < kernels; k++) hipLaunchKernelGGL(spin_kernel, dim3(blocks), dim3(256), 0, main_s, us * 100, sink);
    });
    timed("one stream, hipExtAnyOrderLaunch", [&] {
        for (int k = 0; k < kernels; k++)
            hipExtLaunchKernelGGL(spin_kernel, dim3(blocks), dim3(256), 0, main_s, nullptr, nullptr, k ? hipExtAnyOrderLaunch : 0, us * 100, sink);
    });
    timed("fork / join over side streams", [&] {
        CK(hipEventRecord(fork, main_s));
        for (int k = 1; k < kernels; k++) CK(hipStreamWaitEvent(side[k], fork, 0));
        for (int k = 0; k < kernels; k++) hipLaunchKernelGGL(spin_kernel, dim3(blocks), dim3(256), 0, k ? side[k] : main_s, us * 100, sink);
        for (int k = 1; k < kernels; k++) { CK(hipEventRecord(join[k], side[k])); CK(hipStreamWaitEvent(main_s, join[k], 0)); }
    });
    timed("one kernel of all the blocks", [&] {
        hipLaunchKernelGGL(spin_kernel, dim3(blocks * kernels), dim3(256), 0, main_s, us * 100, sink);
    });
    CK(hipGetLastError());
    return 0;
}
