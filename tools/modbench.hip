// modbench.hip -- store-roof probes for the modulator (diagnostic, not product).
//   ./modbench [n_streams=4096]
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../afskmodem_amd/csrc/afsk_synth.hip"   // the kernels as templates (diagnostic build)
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// plain fill: blocks of 256 threads, ITERS 16-byte stores per thread, block-contiguous
template <int ITERS, bool NT>
__global__ __launch_bounds__(256) void fill_kernel(u32x4* dst, size_t n16, uint32_t v) {
    size_t base = (size_t)blockIdx.x * 256 * ITERS;
#pragma unroll
    for (int it = 0; it < ITERS; it++) {
        size_t i = base + (size_t)it * 256 + threadIdx.x;
        if (i < n16) {
            u32x4 w = {v, v + (uint32_t)it, v, v};
            if (NT) __builtin_nontemporal_store(w, dst + i); else dst[i] = w;
        }
    }
}
// grid-stride persistent fill
template <bool NT>
__global__ __launch_bounds__(256) void fill_gs(u32x4* dst, size_t n16, uint32_t v) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
        u32x4 w = {v, v, v, v};
        if (NT) __builtin_nontemporal_store(w, dst + i); else dst[i] = w;
    }
}

template <class F>
static double time_us(F&& f, int reps = 10) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a, 0));
    for (int k = 0; k < reps; k++) f();
    CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps * 1e3;
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 4096;
    const int L = 48000;
    const size_t bytes = (size_t)n * L * 2, n16 = bytes / 16;
    int16_t* d_x; CK(hipMalloc(&d_x, bytes));
    auto rep = [&](const char* name, double us) { printf("%-34s %8.1f us  %6.2f TB/s\n", name, us, bytes / us * 1e-6); };
    rep("fill 4x16B/thread", time_us([&] { hipLaunchKernelGGL((fill_kernel<4, false>), dim3((n16 + 1023) / 1024), dim3(256), 0, 0, (u32x4*)d_x, n16, 1u); }));
    rep("fill 4x16B/thread nt", time_us([&] { hipLaunchKernelGGL((fill_kernel<4, true>), dim3((n16 + 1023) / 1024), dim3(256), 0, 0, (u32x4*)d_x, n16, 1u); }));
    rep("fill 1x16B/thread", time_us([&] { hipLaunchKernelGGL((fill_kernel<1, false>), dim3((n16 + 255) / 256), dim3(256), 0, 0, (u32x4*)d_x, n16, 1u); }));
    rep("fill 16x16B/thread", time_us([&] { hipLaunchKernelGGL((fill_kernel<16, false>), dim3((n16 + 4095) / 4096), dim3(256), 0, 0, (u32x4*)d_x, n16, 1u); }));
    rep("fill 16x16B/thread nt", time_us([&] { hipLaunchKernelGGL((fill_kernel<16, true>), dim3((n16 + 4095) / 4096), dim3(256), 0, 0, (u32x4*)d_x, n16, 1u); }));
    for (int g : {1024, 2048, 4096, 8192})  {
        char nm[64]; snprintf(nm, 64, "fill grid-stride %d blocks", g);
        rep(nm, time_us([&] { hipLaunchKernelGGL((fill_gs<false>), dim3(g), dim3(256), 0, 0, (u32x4*)d_x, n16, 1u); }));
        snprintf(nm, 64, "fill grid-stride %d blocks nt", g);
        rep(nm, time_us([&] { hipLaunchKernelGGL((fill_gs<true>), dim3(g), dim3(256), 0, 0, (u32x4*)d_x, n16, 1u); }));
    }
    rep("hipMemsetAsync", time_us([&] { CK(hipMemsetAsync(d_x, 1, bytes, 0)); }));

    for (int baud : {1200, 300, 2400}) {
        const int plen_v = baud == 1200 ? 34 : (baud == 300 ? 8 : (baud == 600 ? 16 : 68));
        std::vector<int64_t> off(n); std::vector<int32_t> len(n, L), bf(n, 48000 / baud), pl(n, plen_v), ts(n, baud / 4);
        std::vector<uint8_t> payload((size_t)n * plen_v);
        for (size_t i = 0; i < payload.size(); i++) payload[i] = (uint8_t)((i * 2654435761u) >> 13);
        for (int i = 0; i < n; i++) off[i] = (int64_t)i * L;
        int64_t* d_off; int32_t *d_len, *d_bf, *d_pl, *d_ts; uint8_t* d_payload;
        CK(hipMalloc(&d_off, n * 8)); CK(hipMalloc(&d_len, n * 4));
        CK(hipMalloc(&d_bf, n * 4)); CK(hipMalloc(&d_pl, n * 4)); CK(hipMalloc(&d_ts, n * 4));
        CK(hipMalloc(&d_payload, payload.size()));
        CK(hipMemcpy(d_off, off.data(), n * 8, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_len, len.data(), n * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_bf, bf.data(), n * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_pl, pl.data(), n * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_ts, ts.data(), n * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_payload, payload.data(), payload.size(), hipMemcpyHostToDevice));
        for (int quirk : {1, 0}) {
            afsk::ModulateArgs m{d_payload, plen_v, d_pl, d_bf, d_ts, d_off, d_len, n, quirk, d_x, 0};
            char nm[64];
            typedef int (*mod_fn)(const uint8_t*, int32_t, const int32_t*, const int32_t*, const int32_t*, const int64_t*,
                                  const int32_t*, int32_t, int32_t, int32_t, int16_t*, void*);
            for (const char* path : {(const char*)"../afskmodem_amd/csrc/libafsk_amd.so", (const char*)getenv("MODBENCH_LIB_B")}) {
                if (!path) continue;
                void* h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
                mod_fn f = h ? (mod_fn)dlsym(h, "afsk_modulate_batch") : nullptr;
                if (!f) { printf("cannot load %s\n", path); continue; }
                snprintf(nm, 64, "lib %.18s %d q=%d", strrchr(path, '/') ? strrchr(path, '/') + 1 : path, baud, quirk);
                rep(nm, time_us([&] { f(d_payload, plen_v, d_pl, d_bf, d_ts, d_off, d_len, L, n, quirk, d_x, nullptr); }, 20));
            }
            snprintf(nm, 64, "modulate %d baud quirk=%d iters=2", baud, quirk);
            rep(nm, time_us([&] { CK(afsk::launch_modulate_t<2>(m, L, 0)); }));
            snprintf(nm, 64, "modulate %d baud quirk=%d iters=4", baud, quirk);
            rep(nm, time_us([&] { CK(afsk::launch_modulate_t<4>(m, L, 0)); }));
            snprintf(nm, 64, "modulate %d baud quirk=%d iters=8", baud, quirk);
            rep(nm, time_us([&] { CK(afsk::launch_modulate_t<8>(m, L, 0)); }));
            snprintf(nm, 64, "modulate %d baud quirk=%d iters=12", baud, quirk);
            rep(nm, time_us([&] { CK(afsk::launch_modulate_t<12>(m, L, 0)); }));
            snprintf(nm, 64, "modulate %d q=%d 512thr x4", baud, quirk);
            rep(nm, time_us([&] { CK((afsk::launch_modulate_t<4, 512>(m, L, 0))); }));
            snprintf(nm, 64, "modulate %d q=%d 1024thr x2", baud, quirk);
            rep(nm, time_us([&] { CK((afsk::launch_modulate_t<2, 1024>(m, L, 0))); }));
            snprintf(nm, 64, "modulate %d q=%d 1024thr x1", baud, quirk);
            rep(nm, time_us([&] { CK((afsk::launch_modulate_t<1, 1024>(m, L, 0))); }));
            snprintf(nm, 64, "modulate %d q=%d 512thr x2", baud, quirk);
            rep(nm, time_us([&] { CK((afsk::launch_modulate_t<2, 512>(m, L, 0))); }));
            snprintf(nm, 64, "modulate %d q=%d 128thr x8", baud, quirk);
            rep(nm, time_us([&] { CK((afsk::launch_modulate_t<8, 128>(m, L, 0))); }));
            snprintf(nm, 64, "modulate %d q=%d 64thr x8", baud, quirk);
            rep(nm, time_us([&] { CK((afsk::launch_modulate_t<8, 64>(m, L, 0))); }));
            snprintf(nm, 64, "modulate %d q=%d 64thr x16", baud, quirk);
            rep(nm, time_us([&] { CK((afsk::launch_modulate_t<16, 64>(m, L, 0))); }));
            snprintf(nm, 64, "modulate %d q=%d 128thr x4", baud, quirk);
            rep(nm, time_us([&] { CK((afsk::launch_modulate_t<4, 128>(m, L, 0))); }));
            snprintf(nm, 64, "modulate %d q=%d 128thr x16", baud, quirk);
            rep(nm, time_us([&] { CK((afsk::launch_modulate_t<16, 128>(m, L, 0))); }));
        }
    }
    return 0;
}
