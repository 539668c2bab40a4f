// kbench.hip -- interleaved A/B timing of demod kernel variants in ONE process
// (cdna_hip_programming.md rule 24).  Diagnostic tool, not part of the product.
//   ./kbench [n_streams=4096] [baud=1200] [rounds=15] [reps=5]
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../afskmodem_amd/csrc/afsk_demod_impl.h"
#include "afsk_twopass.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

using afsk::DemodArgs;
typedef void (*launch_fn)(const DemodArgs&, hipStream_t);

static void launch_twopass(const DemodArgs& a, hipStream_t s) {
    const int blocks = (a.n_streams + afsk::kWavesPerBlock - 1) / afsk::kWavesPerBlock;
    hipLaunchKernelGGL((afsk::demod_twopass_kernel_t<0>), dim3(blocks), dim3(64 * afsk::kWavesPerBlock), 0, s, a);
}

// the MIXED-baud kernel compiled into this tool (FLAGS = diagnostics); BIG as launch_demod picks it
#ifdef KBENCH_MIXED
template <int FLAGS>
static void launch_flags(const DemodArgs& a, hipStream_t s) {
    const int blocks = (a.n_streams + afsk::kWavesPerBlock - 1) / afsk::kWavesPerBlock;
    if (a.n_streams >= afsk::kHintMinStreams)
        hipLaunchKernelGGL((afsk::demod_kernel_t<FLAGS, afsk::kWavesPerBlock, 0, true>), dim3(blocks), dim3(64 * afsk::kWavesPerBlock), 0, s, a);
    else
        hipLaunchKernelGGL((afsk::demod_kernel_t<FLAGS, afsk::kWavesPerBlock, 0, false>), dim3(blocks), dim3(64 * afsk::kWavesPerBlock), 0, s, a);
}

#endif

// the UNIFORM kernel of KBENCH_BF (compile-time, default 40 = 1200 baud) with diagnostic FLAGS
#ifndef KBENCH_BF
#define KBENCH_BF 40
#endif
template <int FLAGS>
static void launch_uflags(const DemodArgs& a, hipStream_t s) {
    const int blocks = (a.n_streams + afsk::kWavesPerBlock - 1) / afsk::kWavesPerBlock;
    DemodArgs b = a;
    b.uniform_bit_frames = KBENCH_BF;
    if (a.n_streams >= afsk::uniform_big_from(KBENCH_BF))
        hipLaunchKernelGGL((afsk::demod_uniform_kernel_t<KBENCH_BF, FLAGS, true>), dim3(blocks), dim3(64 * afsk::kWavesPerBlock), 0, s, b);
    else
        hipLaunchKernelGGL((afsk::demod_uniform_kernel_t<KBENCH_BF, FLAGS, false>), dim3(blocks), dim3(64 * afsk::kWavesPerBlock), 0, s, b);
}

// ---- persistent grids around the uniform body (experiments; NOT in the product) ------------------------
// 2 blocks per CU, every wave loops over streams: STEAL = false: s = wave id, += number of waves (static
// striding); STEAL = true: one atomicAdd on a global counter per claimed stream (work stealing).
template <bool STEAL, bool BIG>
__global__ __launch_bounds__(64 * afsk::kWavesPerBlock) void persistent_uniform_kernel(DemodArgs a, unsigned* counter) {
    using namespace afsk;
    __shared__ __attribute__((aligned(16))) uint8_t lds_all[kWavesPerBlock * kFastWaveLdsProduct];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    uint8_t* lds = lds_all + wave * kFastWaveLdsProduct;
    const int total = gridDim.x * kWavesPerBlock;
    int s = blockIdx.x * kWavesPerBlock + wave;
    for (int it = 0; it < (1 << 20); it++) {   // (bounded on top of the exit condition every wave reaches)
        if constexpr (STEAL) {
            unsigned v = 0;
            if (lane == 0) v = atomicAdd(counter, 1u);
            s = (int)__builtin_amdgcn_readfirstlane(v);
        }
        if (s >= a.n_streams) break;
        process_uniform_stream<KBENCH_BF, 0, BIG>(a, s, lds, lane);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        if constexpr (!STEAL) s += total;
    }
}
static unsigned* g_counters = nullptr;       // zeroed once; every launch takes the next one
static int g_counter_next = 0;
template <bool STEAL>
static void launch_persistent(const DemodArgs& a, hipStream_t s) {
    DemodArgs b = a;
    b.uniform_bit_frames = KBENCH_BF;
    const int blocks = std::min((a.n_streams + afsk::kWavesPerBlock - 1) / afsk::kWavesPerBlock, 512);
    unsigned* c = g_counters + (g_counter_next++ & 16383);
    if (a.n_streams >= afsk::kHintMinStreamsUniform)
        hipLaunchKernelGGL((persistent_uniform_kernel<STEAL, true>), dim3(blocks), dim3(64 * afsk::kWavesPerBlock), 0, s, b, c);
    else
        hipLaunchKernelGGL((persistent_uniform_kernel<STEAL, false>), dim3(blocks), dim3(64 * afsk::kWavesPerBlock), 0, s, b, c);
}


// Third persistent form (r3, second session): FIRST item static (wave w takes stream w, no atomic), the rest
// of the launch in 8 queues (queue q = streams total + q + 8 j; one 128-byte-padded counter each, so claims
// do not serialise on one address); a wave claims from the queue of its block's XCD (blocks are dealt
// round-robin over the 8 XCDs) and, once that is exhausted, looks at all eight counters with one 8-lane load
// and steals from the first queue that has work left.  Aim: the hardware deals every XCD the same number of
// blocks although the XCDs are not equally fast (median wave lifetimes differ by ~10 %), so with two items
// per slot the slow XCDs are the kernel's tail; here a fast XCD's waves take a third item from a slow one.
template <bool BIG>
__global__ __launch_bounds__(64 * afsk::kWavesPerBlock) void persistent_xq_kernel(DemodArgs a, unsigned* counters) {
    using namespace afsk;
    __shared__ __attribute__((aligned(16))) uint8_t lds_all[kWavesPerBlock * kFastWaveLdsProduct];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    uint8_t* lds = lds_all + wave * kFastWaveLdsProduct;
    const int total = gridDim.x * kWavesPerBlock;
    int s = blockIdx.x * kWavesPerBlock + wave;
    if (s >= a.n_streams) return;
    process_uniform_stream<KBENCH_BF, 0, BIG>(a, s, lds, lane);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const int rest = a.n_streams - total;                      // streams behind the static first items
    if (rest <= 0) return;
    int q = blockIdx.x & 7;
    for (int it = 0; it < (1 << 20); it++) {                   // (bounded on top of the exit condition every wave reaches)
        const int qlen = (rest - q + 7) >> 3;                  // streams in queue q
        unsigned v = 0;
        if (lane == 0) v = __hip_atomic_fetch_add(counters + 32 * q, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int j = (int)__builtin_amdgcn_readfirstlane(v);
        if (j < qlen) {
            process_uniform_stream<KBENCH_BF, 0, BIG>(a, total + q + 8 * j, lds, lane);
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            continue;
        }
        // own queue exhausted: one look at all eight counters, then the first queue (from q + 1 on) with work left
        unsigned c = 0xffffffffu;
        if (lane < 8) c = __hip_atomic_load(counters + 32 * lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int ql = (rest - (lane & 7) + 7) >> 3;
        const unsigned long long open = __ballot(lane < 8 && (int)c < ql);
        if (open == 0) break;
        const unsigned rot = (unsigned)((open | (open << 8)) >> (q + 1)) & 0xffu;      // bit k = queue (q + 1 + k) & 7
        q = (q + 1 + __builtin_ctz(rot)) & 7;
    }
}
static unsigned* g_counters_xq = nullptr;    // 256 zeroed words per launch (8 counters, 128 bytes apart)
static int g_counter_xq_next = 0;
static void launch_persistent_xq(const DemodArgs& a, hipStream_t s) {
    DemodArgs b = a;
    b.uniform_bit_frames = KBENCH_BF;
    const int blocks = std::min((a.n_streams + afsk::kWavesPerBlock - 1) / afsk::kWavesPerBlock, 512);
    unsigned* c = g_counters_xq + 256 * (g_counter_xq_next++ & 16383);
    if (a.n_streams >= afsk::kHintMinStreamsUniform)
        hipLaunchKernelGGL((persistent_xq_kernel<true>), dim3(blocks), dim3(64 * afsk::kWavesPerBlock), 0, s, b, c);
    else
        hipLaunchKernelGGL((persistent_xq_kernel<false>), dim3(blocks), dim3(64 * afsk::kWavesPerBlock), 0, s, b, c);
}

// same kernel over 4 rotating copies of the input (1.6 GB working set at 4096 streams): defeats
// any reuse of the 256 MiB Infinity Cache between back-to-back launches
static const int16_t* g_copies[4] = {nullptr, nullptr, nullptr, nullptr};
static int g_rot = 0;

// the product library's entry points (same kernels, compiled in their own translation units)
typedef int (*lib_demod_fn)(const int16_t*, const int64_t*, const int32_t*, const int32_t*, int32_t, int32_t,
                            uint8_t*, int32_t, int32_t*, int32_t*, int32_t*, int32_t*, int32_t*, void*);
typedef int (*lib_uniform_fn)(const int16_t*, const int64_t*, const int32_t*, int32_t, int32_t, int32_t,
                              uint8_t*, int32_t, int32_t*, int32_t*, int32_t*, int32_t*, int32_t*, int32_t*, int32_t*,
                              int32_t, void*);
static int g_bf = 40;
// up to five builds of the library: index 0 = ../afskmodem_amd/csrc/libafsk_amd.so, 1.. = KBENCH_LIB_B=<a.so:b.so:...>
static lib_demod_fn g_lib_m[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
static lib_uniform_fn g_lib_u[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
template <int I>
static void launch_lib_m(const DemodArgs& a, hipStream_t s) {
    g_lib_m[I](a.samples, a.stream_offset, a.stream_len, a.bit_frames, a.amp_end, a.n_streams, a.out_bytes,
               a.out_stride, a.out_nbytes, a.out_nbits, a.out_clock_idx, a.out_term_frame, a.out_status, s);
}
template <int I>
static void launch_lib_u(const DemodArgs& a, hipStream_t s) {
    g_lib_u[I](a.samples, a.stream_offset, a.stream_len, g_bf, a.amp_end, a.n_streams, a.out_bytes, a.out_stride,
               a.out_nbytes, a.out_nbits, a.out_clock_idx, a.out_term_frame, a.out_status, nullptr, nullptr, 0, s);
}
template <int I>
static void launch_lib_u_rot(const DemodArgs& a, hipStream_t s) {
    DemodArgs b = a;
    b.samples = g_copies[(g_rot++) & 3];
    launch_lib_u<I>(b, s);
}

struct Variant { const char* name; launch_fn fn; bool exact; };

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 4096;
    const int baud = argc > 2 ? atoi(argv[2]) : 1200;
    const int rounds = argc > 3 ? atoi(argv[3]) : 15;
    const int reps = argc > 4 ? atoi(argv[4]) : 5;
    const int L = 48000, bfv = 48000 / baud;
    const int plen_v = (L - 24000 - 4 * bfv - 4800) / (14 * bfv);   // payload bytes of an (at most) 1 s stream
    std::vector<int64_t> off(n); std::vector<int32_t> len(n, L), bf(n, bfv), pl(n, plen_v), ts(n, baud / 4);
    std::vector<uint8_t> payload((size_t)n * plen_v);
    for (size_t i = 0; i < payload.size(); i++) payload[i] = (uint8_t)((i * 2654435761u) >> 13);
    for (int i = 0; i < n; i++) off[i] = (int64_t)i * L;
    int16_t* d_x; int64_t* d_off; int32_t *d_len, *d_bf, *d_pl, *d_ts; uint8_t* d_payload;
    CK(hipMalloc(&d_x, (size_t)n * L * 2)); CK(hipMalloc(&d_off, n * 8)); CK(hipMalloc(&d_len, n * 4));
    CK(hipMalloc(&d_bf, n * 4)); CK(hipMalloc(&d_pl, n * 4)); CK(hipMalloc(&d_ts, n * 4));
    CK(hipMalloc(&d_payload, payload.size()));
    CK(hipMemcpy(d_off, off.data(), n * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_len, len.data(), n * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_bf, bf.data(), n * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_pl, pl.data(), n * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_ts, ts.data(), n * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_payload, payload.data(), payload.size(), hipMemcpyHostToDevice));
    afsk::ModulateArgs m{d_payload, plen_v, d_pl, d_bf, d_ts, d_off, d_len, n, 1, d_x, 0};
    CK(afsk::launch_modulate(m, L, 0));
    CK(hipDeviceSynchronize());
    {   // modulator timing (write-bound, 2 B per sample)
        hipEvent_t m0, m1; CK(hipEventCreate(&m0)); CK(hipEventCreate(&m1));
        CK(hipEventRecord(m0, 0));
        for (int k = 0; k < 5; k++) CK(afsk::launch_modulate(m, L, 0));
        CK(hipEventRecord(m1, 0)); CK(hipEventSynchronize(m1));
        float ms; CK(hipEventElapsedTime(&ms, m0, m1));
        printf("modulate_kernel: %.1f us per launch, %.2f TB/s written\n", ms / 5 * 1e3, 2.0 * n * L / (ms / 5 * 1e-3) / 1e12);
    }

    {   // live-gate timing (block amplitudes + scan; read-bound, 2 B per sample)
        const int mb = L / 2048, mbursts = 8;
        int32_t *d_amp, *d_nb, *d_bs, *d_bl, *d_oe;
        CK(hipMalloc(&d_amp, (size_t)n * mb * 4)); CK(hipMalloc(&d_nb, n * 4)); CK(hipMalloc(&d_oe, n * 4));
        CK(hipMalloc(&d_bs, (size_t)n * mbursts * 4)); CK(hipMalloc(&d_bl, (size_t)n * mbursts * 4));
        afsk::GateArgs g{d_x, d_off, d_len, 18000, 14000, n, L, mb, mbursts, d_amp, d_nb, d_bs, d_bl, d_oe};   // (max_len = L since r4)
        CK(afsk::launch_gate(g, 0)); CK(hipDeviceSynchronize());
        hipEvent_t m0, m1; CK(hipEventCreate(&m0)); CK(hipEventCreate(&m1));
        CK(hipEventRecord(m0, 0));
        for (int k = 0; k < 5; k++) CK(afsk::launch_gate(g, 0));
        CK(hipEventRecord(m1, 0)); CK(hipEventSynchronize(m1));
        float ms; CK(hipEventElapsedTime(&ms, m0, m1));
        std::vector<int32_t> hn(n); CK(hipMemcpy(hn.data(), d_nb, n * 4, hipMemcpyDeviceToHost));
        long tot = 0; for (int v : hn) tot += v;
        printf("gate kernels: %.1f us per call, %.2f TB/s read (%ld bursts in %d captures)\n", ms / 5 * 1e3,
               2.0 * n * (double)(mb * 2048) / (ms / 5 * 1e-3) / 1e12, tot, n);
    }
    const int stride = 72;
    uint8_t* d_ob; int32_t* d_i32;
    CK(hipMalloc(&d_ob, (size_t)n * stride)); CK(hipMalloc(&d_i32, (size_t)n * 5 * 4));
    DemodArgs a{d_x, d_off, d_len, d_bf, 14000, n, d_ob, stride, d_i32, d_i32 + n, d_i32 + 2 * n, d_i32 + 3 * n, d_i32 + 4 * n};

    // Compiled into this tool: the two-pass baseline and the UNIFORM kernel of KBENCH_BF with its
    // diagnostic variants (-DKBENCH_MIXED adds the mixed-baud kernel: +45 s of compile time).  The product
    // kernels are timed through the library entries (libafsk_amd.so and up to four KBENCH_LIB_B builds).
    g_bf = bfv;
    std::vector<Variant> vs = {
        {"v1 two-pass (r1 baseline)", launch_twopass, true},
#ifdef KBENCH_MIXED
        {"mixed kernel (in-tool)", launch_flags<0>, true},
#endif
    };
#ifndef KBENCH_LITE
    if (bfv == KBENCH_BF) {
        vs.push_back({"uniform kernel (in-tool)", launch_uflags<0>, true});
        vs.push_back({"uniform skip_sync", launch_uflags<1>, true});
        vs.push_back({"uniform skip_valu", launch_uflags<2>, false});
        vs.push_back({"uniform skip_sync+valu", launch_uflags<3>, false});
        CK(hipMalloc(&g_counters, 16384 * sizeof(unsigned))); CK(hipMemset(g_counters, 0, 16384 * sizeof(unsigned)));
        CK(hipMalloc(&g_counters_xq, 256 * 16384 * sizeof(unsigned))); CK(hipMemset(g_counters_xq, 0, 256 * 16384 * sizeof(unsigned)));
        vs.push_back({"persistent, static stride", launch_persistent<false>, true});
        vs.push_back({"persistent, work stealing", launch_persistent<true>, true});
        vs.push_back({"persistent, static first + 8 queues", launch_persistent_xq, true});
    }
    if (bfv == KBENCH_BF)
    {   // timeline of one launch of the diagnostic (stamped) build
        unsigned long long* d_st; CK(hipMalloc(&d_st, (size_t)n * 32)); CK(hipMemset(d_st, 0, (size_t)n * 32));
        DemodArgs as = a; as.debug_stamps = d_st;
        for (int rep = 0; rep < 3; rep++) { launch_uflags<64>(as, 0); CK(hipDeviceSynchronize()); }
        std::vector<unsigned long long> st((size_t)n * 4);
        CK(hipMemcpy(st.data(), d_st, (size_t)n * 32, hipMemcpyDeviceToHost));
        unsigned long long t0 = ~0ull; for (int s = 0; s < n; s++) t0 = std::min(t0, st[4 * s]);
        auto pct = [&](int field, double p) { std::vector<double> v(n); for (int s = 0; s < n; s++) v[s] = (st[4 * s + field] - t0) * 0.01; std::sort(v.begin(), v.end()); return v[(size_t)(p * (n - 1))]; };
        printf("timeline (us since first wave start; min / median / p90 / max over %d streams)\n", n);
        const char* names[4] = {"wave start", "sync done ", "chunk0 in ", "wave end  "};
        for (int f : {0, 2, 1, 3}) printf("  %s %7.2f %7.2f %7.2f %7.2f\n", names[f], pct(f, 0), pct(f, 0.5), pct(f, 0.9), pct(f, 1.0));
        std::vector<double> d0(n), d1(n), d2(n); for (int s = 0; s < n; s++) { d0[s] = (st[4 * s + 2] - st[4 * s]) * 0.01; d1[s] = (st[4 * s + 1] - st[4 * s]) * 0.01; d2[s] = (st[4 * s + 3] - st[4 * s + 1]) * 0.01; }
        std::sort(d0.begin(), d0.end()); std::sort(d1.begin(), d1.end()); std::sort(d2.begin(), d2.end());
        printf("  per-wave: start->first chunk landed median %.2f us (p10 %.2f, p90 %.2f)\n", d0[n / 2], d0[n / 10], d0[9 * n / 10]);
        printf("  per-wave: start->sync done median %.2f us (p10 %.2f, p90 %.2f); sync done->end median %.2f us (p10 %.2f, p90 %.2f)\n",
               d1[n / 2], d1[n / 10], d1[9 * n / 10], d2[n / 2], d2[n / 10], d2[9 * n / 10]);
        // generation split: streams in first half of block ids vs second half
        int late = 0; for (int s = 0; s < n; s++) late += ((st[4 * s] - t0) * 0.01 > 10.0);
        printf("  waves starting later than 10 us after launch: %d of %d\n", late, n);
        // per-XCD view (blocks are dealt round-robin over the 8 XCDs: block b -> group b % 8)
        printf("  wave lifetime by block %% 8 (median us):");
        for (int x = 0; x < 8; x++) {
            std::vector<double> v;
            for (int s = 0; s < n; s++) if (((s / 4) % 8) == x) v.push_back((st[4 * s + 3] - st[4 * s]) * 0.01);
            std::sort(v.begin(), v.end());
            printf(" %.1f", v.empty() ? 0.0 : v[v.size() / 2]);
        }
        printf("\n  kernel span (first start -> last end): %.2f us; last gen-1 end %.2f, first gen-2 start %.2f\n",
               pct(3, 1.0), pct(3, 0.4999), pct(0, 0.5001));
        CK(hipFree(d_st));
    }
#endif
    if (n <= 8192) {
        g_copies[0] = d_x;
        for (int c = 1; c < 4; c++) {
            int16_t* p; CK(hipMalloc(&p, (size_t)n * L * 2));
            CK(hipMemcpy(p, d_x, (size_t)n * L * 2, hipMemcpyDeviceToDevice));
            g_copies[c] = p;
        }
    }
    {
        static const launch_fn tramp_m[5] = {launch_lib_m<0>, launch_lib_m<1>, launch_lib_m<2>, launch_lib_m<3>, launch_lib_m<4>};
        static const launch_fn tramp_u[5] = {launch_lib_u<0>, launch_lib_u<1>, launch_lib_u<2>, launch_lib_u<3>, launch_lib_u<4>};
        static std::string names_m[5], names_u[5];
        std::string list = "../afskmodem_amd/csrc/libafsk_amd.so";
        if (const char* pb = getenv("KBENCH_LIB_B")) list += std::string(":") + pb;
        const bool skip_mixed = getenv("KBENCH_NO_MIXED") != nullptr;
        size_t pos = 0;
        for (int i = 0; i < 5 && pos <= list.size(); i++) {
            size_t e = list.find(':', pos);
            std::string one = list.substr(pos, e == std::string::npos ? std::string::npos : e - pos);
            pos = e == std::string::npos ? list.size() + 1 : e + 1;
            if (one.empty()) continue;
            if (void* h = dlopen(one.c_str(), RTLD_NOW | RTLD_LOCAL)) {
                g_lib_m[i] = (lib_demod_fn)dlsym(h, "afsk_demod_batch");
                g_lib_u[i] = (lib_uniform_fn)dlsym(h, "afsk_demod_batch_uniform");
                size_t sl = one.rfind('/');
                const std::string base = std::string("lib ") + (char)('A' + i) + " " + (sl == std::string::npos ? one : one.substr(sl + 1));
                names_m[i] = base + " mixed";
                names_u[i] = base + " uniform";
                if (g_lib_m[i] && !skip_mixed) vs.push_back(Variant{names_m[i].c_str(), tramp_m[i], true});
                if (g_lib_u[i]) vs.push_back(Variant{names_u[i].c_str(), tramp_u[i], true});
            } else {
                printf("cannot load %s: %s\n", one.c_str(), dlerror());
            }
        }
        if (g_lib_u[0] && g_copies[1]) vs.push_back(Variant{"lib A uniform, 4 rotating inputs", launch_lib_u_rot<0>, true});
    }
    if (const char* only = getenv("KBENCH_ONLY")) {          // run a single variant (cache-state studies)
        std::vector<Variant> keep;
        for (auto& v : vs) if (strstr(v.name, only)) keep.push_back(v);
        if (!keep.empty()) vs = keep;
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<std::vector<float>> times(vs.size());
    // reference outputs from variant 0
    std::vector<uint8_t> ref_ob((size_t)n * stride), ob((size_t)n * stride);
    std::vector<int32_t> ref_i((size_t)n * 5), iv((size_t)n * 5);
    for (size_t v = 0; v < vs.size(); v++) {
        CK(hipMemset(d_ob, 0, (size_t)n * stride)); CK(hipMemset(d_i32, 0xff, (size_t)n * 20));
        vs[v].fn(a, 0); CK(hipDeviceSynchronize());
        CK(hipMemcpy(ob.data(), d_ob, ob.size(), hipMemcpyDeviceToHost));
        CK(hipMemcpy(iv.data(), d_i32, iv.size() * 4, hipMemcpyDeviceToHost));
        if (v == 0) { ref_ob = ob; ref_i = iv;
            long okb = 0; for (int s = 0; s < n; s++) okb += iv[s] == plen_v && !memcmp(&ob[(size_t)s * stride], &payload[(size_t)s * plen_v], plen_v);
            printf("base: %ld / %d streams decode to their payload\n", okb, n);
        } else if (vs[v].exact) {
            printf("%s: outputs %s base\n", vs[v].name, (ob == ref_ob && iv == ref_i) ? "==" : "!=");
        }
    }
    for (int r = 0; r < rounds; r++)
        for (size_t v = 0; v < vs.size(); v++) {
            CK(hipEventRecord(e0, 0));
            for (int k = 0; k < reps; k++) vs[v].fn(a, 0);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (r > 0) times[v].push_back(ms / reps);
        }
    const double bytes = 2.0 * n * L;
    printf("n=%d baud=%d  (full-buffer bytes %.1f MB)\n", n, baud, bytes / 1e6);
    for (size_t v = 0; v < vs.size(); v++) {
        std::sort(times[v].begin(), times[v].end());
        float med = times[v][times[v].size() / 2], mn = times[v][0];
        printf("%-28s median %8.2f us  min %8.2f us   full-buffer %.2f TB/s (median)\n", vs[v].name, med * 1e3, mn * 1e3, bytes / (med * 1e-3) / 1e12);
    }
    return 0;
}
