// ringfloor.hip -- the streaming FLOOR of a work-item granularity (diagnostic, no arithmetic, no library).
//
// The demodulator gives one wavefront one whole stream and streams it through a 16 KiB LDS-DMA ring
// (afsk_demod_fast.h: 16 chunks of 1 KiB requested at wave start, then rounds that consume 5 chunks and
// request 5 more).  Config #2 (4096 streams x 96 KB on 2048 wave slots) is two such work items per slot,
// and the verdict of round 2 asks whether FINER items (a stream split over the 4 waves of a block, or
// over more launch slots) would beat that quantisation.  This tool measures the ceiling of every such
// split without building the decoder around it: each wave streams ONE contiguous piece of P bytes with
// exactly the ring discipline of the product (same builtin, same nt policy, same waits, ds_read_b128 of
// every byte so the data really crosses the LDS), optionally pausing once after its first 8 chunks for
// D microseconds -- the place where the product computes the clock index and has nothing in flight that
// it could consume.  P = 96000 is the product's item; 48000 / 24000 / 12000 are a stream split 2 / 4 / 8
// ways.  Nothing is decoded, so every figure is a lower bound of the corresponding kernel.
//
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o ringfloor ringfloor.hip
//   ./ringfloor [total_MB=393] [reps=200]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// product: 16 chunks of 1 KiB per wave, 5 consumed and re-requested per round (a 1200-baud round is 5 KiB)

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// pause_ticks: s_memrealtime ticks (100 MHz) to idle after the first 8 chunks have landed
// pause_waves: bit w set = wave w of the block pauses (0xF: every wave; 0x1: only wave 0, the others
//              meet it at a workgroup barrier -- the "one wave recovers the clock for the block" form)
template <int WAVES, int kRingChunks, int kRound>
__global__ __launch_bounds__(64 * WAVES) void ring_stream(const char* __restrict__ base, int64_t piece_bytes, int64_t piece_stride,
                                                          int n_pieces, int pause_ticks, int pause_waves,
                                                          int barrier_after_pause, uint32_t* __restrict__ sink) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_all[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int piece = blockIdx.x * WAVES + wv;
    uint8_t* ring = lds_all + wv * (kRingChunks * 1024);
    const bool live = piece < n_pieces;
    const char* src = base + (int64_t)(live ? piece : 0) * piece_stride;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, live ? (int)piece_bytes : 0, 0x00020000);
    const int n_chunks = (int)((piece_bytes + 1023) / 1024);
    int next = 0;
    auto issue = [&](int c) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, LDS_PTR(ring + (c % kRingChunks) * 1024), 16, lane * 16,
                                                 c * 1024, 0, 2 /* nt */);
    };
#pragma unroll
    for (int c = 0; c < kRingChunks; c++) issue(c);      // out-of-range chunks are clipped by the descriptor
    next = kRingChunks;
    wait_vmcnt<kRingChunks - 8>();                        // chunks 0..7 = the sync window
    if (pause_ticks > 0 && ((pause_waves >> wv) & 1)) {
        const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
        while ((int64_t)(__builtin_amdgcn_s_memrealtime() - t0) < pause_ticks) __builtin_amdgcn_s_sleep(4);
    }
    if (barrier_after_pause) __builtin_amdgcn_s_barrier();   // bare s_barrier: __syncthreads() would drain vmcnt first
    uint32_t acc = 0;
    for (int c0 = 0; c0 < n_chunks; c0 += kRound) {
        // chunks c0 .. c0+4 have landed once at most 16 - 5 = 11 requests are outstanding
        wait_vmcnt<kRingChunks - kRound>();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int k = 0; k < kRound; k++) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(ring + ((c0 + k) % kRingChunks) * 1024 + lane * 16);
            acc ^= v[0] ^ v[1] ^ v[2] ^ v[3];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#pragma unroll
        for (int k = 0; k < kRound; k++) issue(next + k);   // beyond the piece: clipped, still counted by vmcnt
        next += kRound;
    }
    wait_vmcnt<0>();
    if (acc == 0x9e3779b9u) sink[0] = acc;                // keeps the reads alive
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int WAVES, int kRingChunks = 16, int kRound = 5>
float run(const std::vector<char*>& bufs, int64_t piece, int64_t stride, int n_pieces, int pause_ticks, int pause_waves, int barrier,
          uint32_t* sink, int reps, hipStream_t st) {
    const int blocks = (n_pieces + WAVES - 1) / WAVES;
    const size_t lds = (size_t)WAVES * kRingChunks * 1024;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int r = 0; r < 20; r++)
        hipLaunchKernelGGL((ring_stream<WAVES, kRingChunks, kRound>), dim3(blocks), dim3(64 * WAVES), lds, st, bufs[r % bufs.size()], piece, stride, n_pieces,
                           pause_ticks, pause_waves, barrier, sink);
    CK(hipEventRecord(e0, st));
    for (int r = 0; r < reps; r++)
        hipLaunchKernelGGL((ring_stream<WAVES, kRingChunks, kRound>), dim3(blocks), dim3(64 * WAVES), lds, st, bufs[r % bufs.size()], piece, stride, n_pieces,
                           pause_ticks, pause_waves, barrier, sink);
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return ms * 1000.0f / reps;
}

int main(int argc, char** argv) {
    const int64_t total = (int64_t)((argc > 1 ? atof(argv[1]) : 393.216) * 1e6);
    const int reps = argc > 2 ? atoi(argv[2]) : 200;
    const int copies = 5;                                  // > 7 x the 256 MiB Infinity Cache in rotation
    std::vector<char*> bufs(copies);
    for (auto& b : bufs) { CK(hipMalloc((void**)&b, (size_t)total + 4096)); CK(hipMemset(b, 0x5a, (size_t)total + 4096)); }
    uint32_t* sink;
    CK(hipMalloc((void**)&sink, 64));
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    // clock settling (bench.py pre-rolls 300 ms for the same reason)
    for (int r = 0; r < 3000; r++)
        hipLaunchKernelGGL((ring_stream<4, 16, 5>), dim3((int)(total / 96000 / 4)), dim3(256), 4 * 16 * 1024, st, bufs[r % copies],
                           (int64_t)96000, (int64_t)96000, (int)(total / 96000), 0, 0, 0, sink);
    CK(hipStreamSynchronize(st));
    printf("ring streaming floor, %.1f MB per launch (%d rotating buffers), %d launches per figure\n", total / 1e6, copies, reps);
    printf("%-64s %9s %9s\n", "work item (per wave), pause", "us", "TB/s");
    struct Case { const char* name; int64_t piece; int pause_us10; int pause_waves; int barrier; int64_t stride = 0; };
    const Case cases[] = {
        {"96000 B (one stream per wave: the product's item), no pause", 96000, 0, 0xF, 0},
        {"96000 B, 3.0 us pause after the sync window (phase A)", 96000, 30, 0xF, 0},
        {"86528 B of every 96000 (what the tail hint leaves to fetch), no pause", 86528, 0, 0xF, 0, 96000},
        {"86528 B of every 96000, 3.0 us pause", 86528, 30, 0xF, 0, 96000},
        {"48000 B (stream split 2 ways), no pause", 48000, 0, 0xF, 0},
        {"48000 B, 3.0 us pause in every wave", 48000, 30, 0xF, 0},
        {"48000 B, 1.5 us pause in every wave (search split 2 ways)", 48000, 15, 0xF, 0},
        {"24000 B (stream split over the 4 waves of a block), no pause", 24000, 0, 0xF, 0},
        {"24000 B, wave 0 pauses 3.0 us, block barrier", 24000, 30, 0x1, 1},
        {"24000 B, every wave pauses 0.8 us (search split 4 ways), barrier", 24000, 8, 0xF, 1},
        {"12000 B (split 8 ways), no pause", 12000, 0, 0xF, 0},
        {"12000 B, every wave pauses 0.8 us", 12000, 8, 0xF, 0},
    };
    for (int pass = 0; pass < 2; pass++)                   // twice, interleaved: boxes drift
        for (const Case& c : cases) {
            const int64_t stride = c.stride ? c.stride : c.piece;
            const int n_pieces = (int)(total / stride);
            const float us = run<4>(bufs, c.piece, stride, n_pieces, c.pause_us10 * 10, c.pause_waves, c.barrier, sink, reps, st);
            printf("%-64s %9.2f %9.3f\n", c.name, us, (double)n_pieces * c.piece / us * 1e-6);
        }
    // ring shapes (96000-byte items, no pause): burst size of the refills, ring depth against waves per CU
    auto shapes = [&](const std::vector<char*>& b, int64_t tot, int r) {
        const int n = (int)(tot / 96000);
        struct R { const char* name; float us; };
        const R rows[] = {
            {"4 waves x 16 KiB ring, rounds of 5 (product)", run<4, 16, 5>(b, 96000, 96000, n, 0, 0, 0, sink, r, st)},
            {"4 waves x 16 KiB ring, rounds of 2", run<4, 16, 2>(b, 96000, 96000, n, 0, 0, 0, sink, r, st)},
            {"4 waves x 16 KiB ring, rounds of 4", run<4, 16, 4>(b, 96000, 96000, n, 0, 0, 0, sink, r, st)},
            {"4 waves x 16 KiB ring, rounds of 8", run<4, 16, 8>(b, 96000, 96000, n, 0, 0, 0, sink, r, st)},
            {"4 waves x 16 KiB ring, rounds of 10", run<4, 16, 10>(b, 96000, 96000, n, 0, 0, 0, sink, r, st)},
            {"3 waves x 20 KiB ring, rounds of 5 (6 waves per CU)", run<3, 20, 5>(b, 96000, 96000, n, 0, 0, 0, sink, r, st)},
            {"3 waves x 20 KiB ring, rounds of 10", run<3, 20, 10>(b, 96000, 96000, n, 0, 0, 0, sink, r, st)},
            {"2 waves x 32 KiB ring, rounds of 10 (4 waves per CU)", run<2, 32, 10>(b, 96000, 96000, n, 0, 0, 0, sink, r, st)},
            {"4 waves x 12 KiB ring, rounds of 4 (3 blocks = 12 waves per CU)", run<4, 12, 4>(b, 96000, 96000, n, 0, 0, 0, sink, r, st)},
            {"4 waves x 10 KiB ring, rounds of 2 (4 blocks = 16 waves per CU)", run<4, 10, 2>(b, 96000, 96000, n, 0, 0, 0, sink, r, st)},
        };
        for (const R& x : rows)
            printf("%-64s %9.2f %9.3f\n", x.name, x.us, (double)n * 96000 / x.us * 1e-6);
    };
    printf("-- ring shapes, %.1f MB per launch --\n", total / 1e6);
    shapes(bufs, total, reps);
    shapes(bufs, total, reps);
    // the same items at 16x the batch (the large-launch regime; one buffer: 6.3 GB cannot be cached)
    for (auto& b : bufs) CK(hipFree(b));
    std::vector<char*> big(1);
    CK(hipMalloc((void**)&big[0], (size_t)total * 16 + 4096));
    CK(hipMemset(big[0], 0x5a, (size_t)total * 16 + 4096));
    printf("-- %.1f MB per launch --\n", total * 16 / 1e6);
    for (int pass = 0; pass < 2; pass++)
        for (const Case& c : cases) {
            if (c.piece == 12000) continue;
            const int64_t stride = c.stride ? c.stride : c.piece;
            const int n_pieces = (int)(total * 16 / stride);
            const float us = run<4>(big, c.piece, stride, n_pieces, c.pause_us10 * 10, c.pause_waves, c.barrier, sink, std::max(4, reps / 10), st);
            printf("%-64s %9.2f %9.3f\n", c.name, us, (double)n_pieces * c.piece / us * 1e-6);
        }
    printf("-- ring shapes, %.1f MB per launch --\n", total * 16 / 1e6);
    shapes(big, total * 16, std::max(4, reps / 10));
    shapes(big, total * 16, std::max(4, reps / 10));
    return 0;
}
