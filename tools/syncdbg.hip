// syncdbg.hip -- dumps per-offset SAD totals of the fast clock recovery and compares them
// with a host brute force (diagnostic).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../afskmodem_amd/csrc/afsk_demod_impl.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(2);} } while (0)

template <int BF>
__global__ __launch_bounds__(64) void k_sync(const int16_t* xs, int32_t len, uint32_t* dbg, int* ci_out) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[afsk::kFastWaveLdsProduct];
    afsk::FastRing fr;
    fr.rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)xs, 0, len * 2, 0x00020000);
    fr.ring = lds; fr.lane = threadIdx.x;
    for (int c = 0; c < afsk::kRingChunks; c++) fr.issue(c);
    fr.next = afsk::kRingChunks;
    int ci;
    if constexpr (BF <= 120) ci = afsk::recover_clock_index_lanes<BF, true>(fr, dbg);          // contiguous lane windows
    else ci = afsk::recover_clock_index_lane_steps<BF, true>(fr, dbg);                        // sub-windows in steps
    afsk::wait_vmcnt<0>();
    if (threadIdx.x == 0) *ci_out = ci;
}

template <int BF> void run(const std::vector<int16_t>& x) {
    int16_t* d; uint32_t* dbg; int* dci;
    CK(hipMalloc(&d, x.size() * 2)); CK(hipMalloc(&dbg, 4096 * 4 + 256)); CK(hipMalloc(&dci, 4));
    CK(hipMemcpy(d, x.data(), x.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemset(dbg, 0, 4096 * 4 + 256));
    k_sync<BF><<<1, 64>>>(d, (int)x.size(), dbg, dci);
    CK(hipDeviceSynchronize());
    std::vector<uint32_t> h(4096 + 64); int ci;
    CK(hipMemcpy(h.data(), dbg, 4096 * 4 + 256, hipMemcpyDeviceToHost)); CK(hipMemcpy(&ci, dci, 4, hipMemcpyDeviceToHost));
    const int N = 2 * BF, Q = BF / 4, H = BF / 2, NOFF = 4096 - N;
    int bad = 0, first = -1; long best = -1; int bi = 0;
    for (int i = 0; i < NOFF; i++) {
        long tot = 0;
        for (int j = 0; j < N; j++) {
            int t = j < BF ? (((j / Q) & 1) ? -32768 : 32767) : ((j - BF) < H ? 32767 : -32768);
            long dd = t - x[i + j]; tot += dd < 0 ? -dd : dd;
        }
        long mean = tot / N;
        if (best < 0 || mean < best) { best = mean; bi = i; }
        if ((long)h[i] != tot) { if (first < 0) first = i; bad++; }
    }
    printf("BF=%d: gpu ci=%d host ci=%d, totals differing: %d (first at %d: gpu %u)\n", BF, ci, bi, bad, first, first >= 0 ? h[first] : 0);
    if (first >= 0) { printf("  bad offsets:"); int c = 0; for (int i = 0; i < NOFF && c < 40; i++) { long tot = 0; for (int j = 0; j < N; j++) { int t = j < BF ? (((j / Q) & 1) ? -32768 : 32767) : ((j - BF) < H ? 32767 : -32768); long dd = t - x[i + j]; tot += dd < 0 ? -dd : dd; } if ((long)h[i] != tot) { printf(" %d", i); c++; } } printf("\n"); }
}

int main() {
    std::vector<int16_t> x(6000);
    unsigned s = 12345;
    for (auto& v : x) { s = s * 1664525u + 1013904223u; v = (int16_t)(s >> 16); }
    run<20>(x); run<40>(x); run<80>(x); run<160>(x);
    run<100>(x); run<120>(x); run<240>(x); run<300>(x); run<320>(x); run<480>(x); run<500>(x); run<1500>(x); run<2000>(x);   // long / odd quarter lengths
    // a Transmitter-like stream: clean training cycles (many equal minima, first one wins)
    for (size_t i = 0; i < x.size(); i++) { int ph = (int)(i % 80); x[i] = ph < 40 ? (((ph / 10) & 1) ? -32768 : 32767) : (ph < 60 ? 32767 : -32768); }
    run<40>(x);
    for (size_t i = 0; i < x.size(); i++) { int ph = (int)((i + 13) % 40); x[i] = (int16_t)((ph < 20 ? (((ph / 5) & 1) ? -30000 : 30000) : (ph < 30 ? 30000 : -30000)) + (int)(i % 7) - 3); }
    run<20>(x);
    return 0;
}
