// Hardware probes for the access forms the demod kernel wants to rely on
// (gfx950). Each probe runs in its own process: ./hw_probe <name>
//   galign   : global_load_dwordx4 from 2-byte-aligned addresses
//   gldslds  : global_load_lds_dwordx4 from 2-byte-aligned source addresses
//   bufalign : raw_buffer_load_b128 with a 2-byte-aligned SRD base + range check detail
//   buflds   : raw_buffer_load_lds (16 B) with a 2-byte-aligned SRD base + range check
//   dsalign  : ds_read_b32/b64/b128 at 2-byte-aligned LDS addresses
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#define CK(x)                                                                    \
    do {                                                                         \
        hipError_t e_ = (x);                                                     \
        if (e_ != hipSuccess) {                                                  \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            return 2;                                                            \
        }                                                                        \
    } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(2))) u128p { uint32_t v[4]; };

__global__ void k_galign(const int16_t* src, int shift, uint32_t* out) {
    const u128p* p = reinterpret_cast<const u128p*>(src + shift + 8 * threadIdx.x);
    u128p v = *p;
    for (int j = 0; j < 4; j++) out[4 * threadIdx.x + j] = v.v[j];
}

__global__ void k_gldslds(const int16_t* src, int shift, uint32_t* out) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[256];
    const char* g = reinterpret_cast<const char*>(src + shift) + 16 * threadIdx.x;
    __builtin_amdgcn_global_load_lds(
        (const __attribute__((address_space(1))) void*)g,
        (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int j = 0; j < 4; j++) out[4 * threadIdx.x + j] = lds[4 * threadIdx.x + j];
}

__global__ void k_bufalign(const int16_t* src, int shift, int nbytes, uint32_t* out) {
    auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(src + shift), 0, nbytes, 0x00020000);
    u32x4 v = __builtin_bit_cast(
        u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, 16 * threadIdx.x, 0, 0));
    for (int j = 0; j < 4; j++) out[4 * threadIdx.x + j] = v[j];
}

__global__ void k_buflds(const int16_t* src, int shift, int nbytes, uint32_t* out) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[256];
    for (int j = 0; j < 4; j++) lds[4 * threadIdx.x + j] = 0xDEADBEEFu;
    __syncthreads();
    auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(src + shift), 0, nbytes, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds, 16,
                                         16 * threadIdx.x, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int j = 0; j < 4; j++) out[4 * threadIdx.x + j] = lds[4 * threadIdx.x + j];
}

__global__ void k_dsalign(const int16_t* src, int shift, uint32_t* out32, uint32_t* out64,
                          uint32_t* out128) {
    __shared__ __attribute__((aligned(16))) int16_t lds[1024];
    for (int j = threadIdx.x; j < 1024; j += 64) lds[j] = src[j];
    __syncthreads();
    uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) int16_t*)lds;
    uint32_t a32 = base + 2 * shift + 4 * threadIdx.x;
    uint32_t a64 = base + 2 * shift + 8 * threadIdx.x;
    uint32_t a128 = base + 2 * shift + 16 * threadIdx.x;
    uint32_t r32;
    uint64_t r64;
    u32x4 r128;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(r32) : "v"(a32) : "memory");
    asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(r64) : "v"(a64) : "memory");
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(r128) : "v"(a128) : "memory");
    out32[threadIdx.x] = r32;
    out64[2 * threadIdx.x] = (uint32_t)r64;
    out64[2 * threadIdx.x + 1] = (uint32_t)(r64 >> 32);
    for (int j = 0; j < 4; j++) out128[4 * threadIdx.x + j] = r128[j];
}

static uint32_t expect_dword(const std::vector<int16_t>& h, long sample_idx) {
    uint16_t lo = (uint16_t)h[sample_idx], hi = (uint16_t)h[sample_idx + 1];
    return (uint32_t)lo | ((uint32_t)hi << 16);
}

int main(int argc, char** argv) {
    if (argc < 2) { printf("usage: hw_probe <probe>\n"); return 1; }
    const char* name = argv[1];
    const int N = 4096;
    std::vector<int16_t> h(N);
    for (int i = 0; i < N; i++) h[i] = (int16_t)(i * 7 + 1);
    int16_t* d;
    uint32_t *o, *o2, *o3;
    CK(hipMalloc(&d, N * 2));
    CK(hipMalloc(&o, 4096 * 4));
    CK(hipMalloc(&o2, 4096 * 4));
    CK(hipMalloc(&o3, 4096 * 4));
    CK(hipMemcpy(d, h.data(), N * 2, hipMemcpyHostToDevice));
    std::vector<uint32_t> r(4096), r2(4096), r3(4096);
    int bad_total = 0;

    if (!strcmp(name, "galign") || !strcmp(name, "gldslds")) {
        for (int shift = 0; shift < 9; shift++) {
            CK(hipMemset(o, 0, 4096 * 4));
            if (!strcmp(name, "galign")) k_galign<<<1, 64>>>(d, shift, o);
            else k_gldslds<<<1, 64>>>(d, shift, o);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(r.data(), o, 256 * 4, hipMemcpyDeviceToHost));
            int bad = 0;
            for (int i = 0; i < 256; i++) bad += r[i] != expect_dword(h, shift + 2 * i);
            printf("%s shift=%d bad=%d first=%08x expect=%08x\n", name, shift, bad, r[0],
                   expect_dword(h, shift));
            bad_total += bad;
        }
    } else if (!strcmp(name, "bufalign") || !strcmp(name, "buflds")) {
        for (int shift = 0; shift < 4; shift++) {
            for (int nbytes : {1024, 1000, 998, 1002, 20, 18}) {
                CK(hipMemset(o, 0, 4096 * 4));
                if (!strcmp(name, "bufalign")) k_bufalign<<<1, 64>>>(d, shift, nbytes, o);
                else k_buflds<<<1, 64>>>(d, shift, nbytes, o);
                CK(hipDeviceSynchronize());
                CK(hipMemcpy(r.data(), o, 256 * 4, hipMemcpyDeviceToHost));
                // classify each dword: exact / zero / other
                int exact = 0, zero = 0, other = 0, first_nonexact = -1;
                for (int i = 0; i < 256; i++) {
                    uint32_t e = expect_dword(h, shift + 2 * i);
                    if (r[i] == e) exact++;
                    else {
                        if (first_nonexact < 0) first_nonexact = i;
                        if (r[i] == 0) zero++; else other++;
                    }
                }
                printf("%s shift=%d nbytes=%d exact=%d zero=%d other=%d first_nonexact_dword=%d",
                       name, shift, nbytes, exact, zero, other, first_nonexact);
                if (first_nonexact >= 0)
                    printf(" val=%08x expect=%08x", r[first_nonexact],
                           expect_dword(h, shift + 2 * first_nonexact));
                printf("\n");
                int full = nbytes / 4; if (full > 256) full = 256;
                if (first_nonexact >= 0 && first_nonexact < full) bad_total++;
            }
        }
    } else if (!strcmp(name, "dsalign")) {
        for (int shift = 0; shift < 9; shift++) {
            k_dsalign<<<1, 64>>>(d, shift, o, o2, o3);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(r.data(), o, 64 * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(r2.data(), o2, 128 * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(r3.data(), o3, 256 * 4, hipMemcpyDeviceToHost));
            int b32 = 0, b64 = 0, b128 = 0;
            for (int i = 0; i < 64; i++) b32 += r[i] != expect_dword(h, shift + 2 * i);
            for (int i = 0; i < 128; i++) b64 += r2[i] != expect_dword(h, shift + 2 * i);
            for (int i = 0; i < 256; i++) b128 += r3[i] != expect_dword(h, shift + 2 * i);
            printf("dsalign shift=%d bad32=%d bad64=%d bad128=%d (first128=%08x expect=%08x)\n",
                   shift, b32, b64, b128, r3[0], expect_dword(h, shift));
            bad_total += b32 + b64 + b128;
        }
    } else {
        printf("unknown probe\n");
        return 1;
    }
    printf("%s: %s\n", name, bad_total == 0 ? "ALL_OK" : "SOME_MISMATCH");
    return 0;
}
