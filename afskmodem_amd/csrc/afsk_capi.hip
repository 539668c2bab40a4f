// afsk_capi.hip -- extern "C" boundary of libafsk_amd.so (see include/afsk_amd.h).
// Host-side argument checks, error strings and kernel launches; no torch types.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/afsk_amd.h"
#include "afsk_kernels.h"

namespace {

thread_local std::string g_last_error;

int fail(int code, const std::string& msg) {
    g_last_error = msg;
    return code;
}

int hip_fail(hipError_t e, const char* what) {
    return fail(AFSK_E_HIP, std::string(what) + ": " + hipGetErrorString(e));
}

int require_device() {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        return fail(AFSK_E_NO_DEVICE, "no HIP device visible: libafsk_amd has no CPU fallback");
    }
    return AFSK_OK;
}

#define AFSK_HIP(call, what)                             \
    do {                                                 \
        hipError_t e_ = (call);                          \
        if (e_ != hipSuccess) { rc = hip_fail(e_, what); goto done; } \
    } while (0)

}  // namespace

extern "C" {

int afsk_version(void) { return AFSK_ABI_VERSION; }

int afsk_last_error(char* buf, int cap) {
    if (buf && cap > 0) {
        std::strncpy(buf, g_last_error.c_str(), (size_t)cap - 1);
        buf[cap - 1] = '\0';
    }
    return (int)g_last_error.size();
}

int afsk_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

int afsk_sync(void* hip_stream) {
    if (int rc = require_device()) return rc;
    hipError_t e = hipStreamSynchronize((hipStream_t)hip_stream);
    return e == hipSuccess ? AFSK_OK : hip_fail(e, "hipStreamSynchronize");
}

int afsk_demod_batch(const int16_t* samples, const int64_t* stream_offset,
                     const int32_t* stream_len, const int32_t* bit_frames,
                     int32_t amp_end_threshold, int32_t n_streams, uint8_t* out_bytes,
                     int32_t out_stride, int32_t* out_nbytes, int32_t* out_nbits,
                     int32_t* out_clock_idx, int32_t* out_term_frame, int32_t* out_status,
                     void* hip_stream) {
    return afsk_demod_batch_ex(samples, stream_offset, stream_len, bit_frames, amp_end_threshold,
                               n_streams, out_bytes, out_stride, out_nbytes, out_nbits,
                               out_clock_idx, out_term_frame, out_status, nullptr, nullptr, 0,
                               hip_stream);
}

int afsk_demod_batch_ex(const int16_t* samples, const int64_t* stream_offset,
                        const int32_t* stream_len, const int32_t* bit_frames,
                        int32_t amp_end_threshold, int32_t n_streams, uint8_t* out_bytes,
                        int32_t out_stride, int32_t* out_nbytes, int32_t* out_nbits,
                        int32_t* out_clock_idx, int32_t* out_term_frame, int32_t* out_status,
                        int32_t* out_corrected, int32_t* out_margins, int32_t margin_stride,
                        void* hip_stream) {
    if (n_streams < 0 || out_stride < 0 || margin_stride < 0)
        return fail(AFSK_E_INVALID_ARG, "negative size");
    if (n_streams == 0) return AFSK_OK;
    if (!samples || !stream_offset || !stream_len || !bit_frames || !out_nbytes || !out_nbits ||
        !out_clock_idx || !out_term_frame || !out_status || (!out_bytes && out_stride > 0))
        return fail(AFSK_E_INVALID_ARG, "null pointer argument");
    if (int rc = require_device()) return rc;
    afsk::DemodArgs a;
    a.samples = samples; a.stream_offset = stream_offset; a.stream_len = stream_len;
    a.bit_frames = bit_frames; a.amp_end = amp_end_threshold; a.n_streams = n_streams;
    a.out_bytes = out_bytes; a.out_stride = out_stride; a.out_nbytes = out_nbytes;
    a.out_nbits = out_nbits; a.out_clock_idx = out_clock_idx; a.out_term_frame = out_term_frame;
    a.out_status = out_status;
    a.out_corrected = out_corrected;
    a.out_margins = margin_stride > 0 ? out_margins : nullptr;
    a.margin_stride = margin_stride;
    hipError_t e = afsk::launch_demod(a, (hipStream_t)hip_stream);
    return e == hipSuccess ? AFSK_OK : hip_fail(e, "launch demod_kernel");
}

int afsk_demod_batch_host(const int16_t* samples, int64_t total_samples,
                          const int64_t* stream_offset, const int32_t* stream_len,
                          const int32_t* bit_frames, int32_t amp_end_threshold,
                          int32_t n_streams, uint8_t* out_bytes, int32_t out_stride,
                          int32_t* out_nbytes, int32_t* out_nbits, int32_t* out_clock_idx,
                          int32_t* out_term_frame, int32_t* out_status) {
    if (n_streams < 0 || out_stride < 0 || total_samples < 0)
        return fail(AFSK_E_INVALID_ARG, "negative size");
    if (n_streams == 0) return AFSK_OK;
    if (!stream_offset || !stream_len || !bit_frames || !out_nbytes || !out_nbits ||
        !out_clock_idx || !out_term_frame || !out_status || (!out_bytes && out_stride > 0) ||
        (!samples && total_samples > 0))
        return fail(AFSK_E_INVALID_ARG, "null pointer argument");
    for (int32_t s = 0; s < n_streams; s++) {
        const int bf = bit_frames[s];
        if (bf < 4 || (bf & 3) || 2 * bf >= AFSK_SYNC_WINDOW)
            return fail(AFSK_E_INVALID_BAUD, "bit_frames must be a multiple of 4 with 2*bf < 4096");
        if (stream_len[s] < 0 || stream_offset[s] < 0 ||
            stream_offset[s] + stream_len[s] > total_samples)
            return fail(AFSK_E_INVALID_ARG, "stream outside the sample buffer");
    }
    if (int rc0 = require_device()) return rc0;

    int rc = AFSK_OK;
    const size_t n = (size_t)n_streams;
    const size_t sample_bytes = (size_t)(total_samples > 0 ? total_samples : 1) * 2;
    const size_t bytes_out = n * (size_t)(out_stride > 0 ? out_stride : 1);
    char* d_all = nullptr;
    // one allocation: samples | offsets | len | bf | 5 x int32 outputs | bytes
    size_t o_samples = 0;
    size_t o_off = (sample_bytes + 15) & ~(size_t)15;
    size_t o_len = o_off + n * 8;
    size_t o_bf = o_len + n * 4;
    size_t o_i32 = o_bf + n * 4;
    size_t o_bytes = o_i32 + 5 * n * 4;
    size_t total = o_bytes + bytes_out;
    hipStream_t stream = nullptr;
    AFSK_HIP(hipMalloc((void**)&d_all, total), "hipMalloc");
    if (total_samples > 0)
        AFSK_HIP(hipMemcpyAsync(d_all + o_samples, samples, (size_t)total_samples * 2,
                                hipMemcpyHostToDevice, stream), "H2D samples");
    AFSK_HIP(hipMemcpyAsync(d_all + o_off, stream_offset, n * 8, hipMemcpyHostToDevice, stream), "H2D offsets");
    AFSK_HIP(hipMemcpyAsync(d_all + o_len, stream_len, n * 4, hipMemcpyHostToDevice, stream), "H2D lengths");
    AFSK_HIP(hipMemcpyAsync(d_all + o_bf, bit_frames, n * 4, hipMemcpyHostToDevice, stream), "H2D bit_frames");
    {
        int32_t* i32 = (int32_t*)(d_all + o_i32);
        rc = afsk_demod_batch((const int16_t*)(d_all + o_samples), (const int64_t*)(d_all + o_off),
                              (const int32_t*)(d_all + o_len), (const int32_t*)(d_all + o_bf),
                              amp_end_threshold, n_streams, (uint8_t*)(d_all + o_bytes), out_stride,
                              i32, i32 + n, i32 + 2 * n, i32 + 3 * n, i32 + 4 * n, stream);
        if (rc != AFSK_OK) goto done;
        AFSK_HIP(hipMemcpyAsync(out_nbytes, i32, n * 4, hipMemcpyDeviceToHost, stream), "D2H");
        AFSK_HIP(hipMemcpyAsync(out_nbits, i32 + n, n * 4, hipMemcpyDeviceToHost, stream), "D2H");
        AFSK_HIP(hipMemcpyAsync(out_clock_idx, i32 + 2 * n, n * 4, hipMemcpyDeviceToHost, stream), "D2H");
        AFSK_HIP(hipMemcpyAsync(out_term_frame, i32 + 3 * n, n * 4, hipMemcpyDeviceToHost, stream), "D2H");
        AFSK_HIP(hipMemcpyAsync(out_status, i32 + 4 * n, n * 4, hipMemcpyDeviceToHost, stream), "D2H");
        if (out_stride > 0)
            AFSK_HIP(hipMemcpyAsync(out_bytes, d_all + o_bytes, n * (size_t)out_stride,
                                    hipMemcpyDeviceToHost, stream), "D2H bytes");
    }
    AFSK_HIP(hipStreamSynchronize(stream), "hipStreamSynchronize");
done:
    if (d_all) (void)hipFree(d_all);
    return rc;
}

int afsk_modulate_batch(const uint8_t* payload, int32_t payload_stride,
                        const int32_t* payload_len, const int32_t* bit_frames,
                        const int32_t* ts_cycles, const int64_t* stream_offset,
                        const int32_t* stream_len, int32_t max_stream_len, int32_t n_streams,
                        int32_t wav_quirk, int16_t* samples, void* hip_stream) {
    if (n_streams < 0 || payload_stride < 0 || max_stream_len < 0)
        return fail(AFSK_E_INVALID_ARG, "negative size");
    if (n_streams == 0 || max_stream_len == 0) return AFSK_OK;
    if (!payload_len || !bit_frames || !ts_cycles || !stream_offset || !stream_len || !samples ||
        (!payload && payload_stride > 0))
        return fail(AFSK_E_INVALID_ARG, "null pointer argument");
    if (int rc = require_device()) return rc;
    afsk::ModulateArgs a;
    a.payload = payload; a.payload_stride = payload_stride; a.payload_len = payload_len;
    a.bit_frames = bit_frames; a.ts_cycles = ts_cycles; a.stream_offset = stream_offset;
    a.stream_len = stream_len; a.n_streams = n_streams; a.wav_quirk = wav_quirk;
    a.samples = samples; a.chunks = 0;
    hipError_t e = afsk::launch_modulate(a, max_stream_len, (hipStream_t)hip_stream);
    return e == hipSuccess ? AFSK_OK : hip_fail(e, "launch modulate_kernel");
}

int afsk_gate_batch(const int16_t* samples, const int64_t* stream_offset,
                    const int32_t* stream_len, int32_t max_stream_len,
                    int32_t amp_start_threshold, int32_t amp_end_threshold, int32_t n_streams,
                    int32_t max_bursts, int32_t* block_amp, int32_t* out_n_bursts,
                    int32_t* out_burst_start, int32_t* out_burst_len, int32_t* out_open_end,
                    void* hip_stream) {
    if (n_streams < 0 || max_stream_len < 0 || max_bursts < 0)
        return fail(AFSK_E_INVALID_ARG, "negative size");
    if (n_streams == 0) return AFSK_OK;
    const int32_t max_blocks = max_stream_len / 2048;
    if (!samples || !stream_offset || !stream_len || !out_n_bursts || !out_open_end ||
        (max_blocks > 0 && !block_amp) || (max_bursts > 0 && (!out_burst_start || !out_burst_len)))
        return fail(AFSK_E_INVALID_ARG, "null pointer argument");
    if (int rc = require_device()) return rc;
    afsk::GateArgs a;
    a.samples = samples; a.stream_offset = stream_offset; a.stream_len = stream_len;
    a.amp_start = amp_start_threshold; a.amp_end = amp_end_threshold; a.n_streams = n_streams;
    a.max_blocks = max_blocks; a.max_bursts = max_bursts; a.block_amp = block_amp;
    a.out_n_bursts = out_n_bursts; a.out_burst_start = out_burst_start;
    a.out_burst_len = out_burst_len; a.out_open_end = out_open_end;
    hipError_t e = afsk::launch_gate(a, (hipStream_t)hip_stream);
    return e == hipSuccess ? AFSK_OK : hip_fail(e, "launch gate kernels");
}

int afsk_add_noise_batch(int16_t* samples, const int64_t* stream_offset,
                         const int32_t* stream_len, int32_t max_stream_len,
                         const int32_t* scale_q24, int32_t n_streams, uint32_t seed,
                         uint32_t stream_idx_base, void* hip_stream) {
    if (n_streams < 0 || max_stream_len < 0) return fail(AFSK_E_INVALID_ARG, "negative size");
    if (n_streams == 0 || max_stream_len == 0) return AFSK_OK;
    if (!samples || !stream_offset || !stream_len || !scale_q24)
        return fail(AFSK_E_INVALID_ARG, "null pointer argument");
    if (int rc = require_device()) return rc;
    afsk::NoiseArgs a;
    a.samples = samples; a.stream_offset = stream_offset; a.stream_len = stream_len;
    a.scale_q24 = scale_q24; a.n_streams = n_streams; a.seed = seed;
    a.stream_idx_base = stream_idx_base; a.chunks = 0;
    hipError_t e = afsk::launch_noise(a, max_stream_len, (hipStream_t)hip_stream);
    return e == hipSuccess ? AFSK_OK : hip_fail(e, "launch noise_kernel");
}

}  // extern "C"
