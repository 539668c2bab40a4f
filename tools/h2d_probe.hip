// h2d_probe.hip -- host-to-device transfer options for the host-buffer entry (diagnostic).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
    const size_t mb = argc > 1 ? atoi(argv[1]) : 393;
    const size_t bytes = mb << 20;
    char* h = (char*)malloc(bytes);
    memset(h, 1, bytes);
    char* d; CK(hipMalloc(&d, bytes));
    CK(hipDeviceSynchronize());
    auto rate = [&](const char* name, double s) { printf("%-52s %8.2f ms  %6.2f GB/s\n", name, s * 1e3, bytes / s / 1e9); };
    for (int rep = 0; rep < 2; rep++) {
        double t = now(); CK(hipMemcpy(d, h, bytes, hipMemcpyHostToDevice)); rate("hipMemcpy pageable", now() - t);
    }
    {
        double t = now(); CK(hipHostRegister(h, bytes, hipHostRegisterDefault)); double tr = now() - t;
        rate("hipHostRegister (cost)", tr);
        t = now(); CK(hipMemcpy(d, h, bytes, hipMemcpyHostToDevice)); rate("hipMemcpy from registered", now() - t);
        t = now(); CK(hipMemcpy(d, h, bytes, hipMemcpyHostToDevice)); rate("hipMemcpy from registered (2nd)", now() - t);
        t = now(); CK(hipHostUnregister(h)); rate("hipHostUnregister (cost)", now() - t);
    }
    {
        char* p; double t = now(); CK(hipHostMalloc(&p, bytes, hipHostMallocDefault)); rate("hipHostMalloc (cost)", now() - t);
        t = now(); memcpy(p, h, bytes); rate("memcpy pageable -> pinned, 1 thread", now() - t);
        for (int nt : {4, 8, 16}) {
            t = now();
            std::vector<std::thread> th;
            for (int i = 0; i < nt; i++) th.emplace_back([&, i] { size_t lo = bytes * i / nt, hi = bytes * (i + 1) / nt; memcpy(p + lo, h + lo, hi - lo); });
            for (auto& x : th) x.join();
            char nm[64]; snprintf(nm, 64, "memcpy pageable -> pinned, %d threads", nt); rate(nm, now() - t);
        }
        t = now(); CK(hipMemcpy(d, p, bytes, hipMemcpyHostToDevice)); rate("hipMemcpy pinned", now() - t);
        // chunked pipeline: memcpy chunk k+1 into staging while chunk k is in flight
        const size_t chunk = 16 << 20;
        hipStream_t s; CK(hipStreamCreate(&s));
        hipEvent_t ev[2]; CK(hipEventCreate(&ev[0])); CK(hipEventCreate(&ev[1]));
        t = now();
        for (size_t o = 0, k = 0; o < bytes; o += chunk, k++) {
            size_t n = bytes - o < chunk ? bytes - o : chunk;
            char* st = p + (k & 1) * chunk;
            if (k >= 2) CK(hipEventSynchronize(ev[k & 1]));
            memcpy(st, h + o, n);
            CK(hipMemcpyAsync(d + o, st, n, hipMemcpyHostToDevice, s));
            CK(hipEventRecord(ev[k & 1], s));
        }
        CK(hipStreamSynchronize(s));
        rate("pipelined 16 MiB staging, 1 copy thread", now() - t);
        t = now(); CK(hipMemcpy(h, d, 1 << 20, hipMemcpyDeviceToHost)); printf("D2H 1 MiB pageable: %.3f ms\n", (now() - t) * 1e3);
        t = now(); CK(hipHostFree(p)); rate("hipHostFree (cost)", now() - t);
    }
    {
        double t = now(); char* d2; CK(hipMalloc(&d2, bytes)); printf("hipMalloc %zu MB: %.3f ms\n", mb, (now() - t) * 1e3);
        t = now(); CK(hipFree(d2)); printf("hipFree: %.3f ms\n", (now() - t) * 1e3);
    }
    return 0;
}
