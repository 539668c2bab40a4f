// afsk_gate.hip -- batched replay of Receiver.__listen (reference afskmodem.py:299-319): the
// 2048-frame block amplitude gate that cuts bursts out of a continuous capture before they are
// demodulated.  Two kernels:
//
//   block_amp_kernel  one wavefront per 2048-sample block: four coalesced 16-byte loads per
//                     lane, v_sad_u16 accumulation of |x| (ref:94-98), DPP wave reduction,
//                     amp = sum >> 11 (= int(sum / 2048)).  HBM-bound, 2 B per sample.
//   gate_scan_kernel  one thread per capture walks its block amplitudes through the
//                     listen state machine: discard one block (ref:303), wait for
//                     amp > amp_start (ref:306), record through the first amp < amp_end
//                     (ref:316), repeat for the next receive() call; optionally (r6) it also lays the
//                     bursts out as fixed demodulator slots (absolute offset + length per slot).
#include "afsk_kernels.h"

namespace afsk {

constexpr int kListenBlock = 2048;   // ref:189, 209, 310

// 16 bytes from a 2-byte-aligned address (gfx950 global loads are alignment-agnostic), read
// with the non-temporal policy: every byte of a capture is read exactly once.
typedef uint32_t vec16 __attribute__((ext_vector_type(4), aligned(2)));

__global__ __launch_bounds__(256) void block_amp_kernel(GateArgs a) {
    const int lane = threadIdx.x & 63;
    // (xcd_block: the 32 amplitudes of one 128-byte line are written by 8 workgroups -- let them sit on ONE XCD, so that
    // the line leaves its L2 once, whole: afsk_kernels.h)
    const int64_t w = (int64_t)xcd_block((int)blockIdx.x, (int)gridDim.x) * 4 + (threadIdx.x >> 6);   // global wave = block slot
    const int s = (int)(w / a.max_blocks);
    if (s >= a.n_streams) return;
    const int b = (int)(w - (int64_t)s * a.max_blocks);
    const int32_t len = a.stream_len[s];
    // a capture longer than the caller's own bound (max_stream_len, which sized the workspace rows and the
    // grid) or negative is refused, not read: device-side lengths are data the host never saw
    if ((uint32_t)len > (uint32_t)a.max_len) return;
    const int32_t nb = len / kListenBlock;
    if (b >= nb) return;
    const int16_t* src = a.samples + a.stream_offset[s] + (int64_t)b * kListenBlock;
    uint32_t acc = 0;
    vec16 v[4];
#pragma unroll
    for (int j = 0; j < 4; j++)
        v[j] = __builtin_nontemporal_load(reinterpret_cast<const vec16*>(src + 512 * j + 8 * lane));
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
        for (int k = 0; k < 4; k++)
            acc = __builtin_amdgcn_sad_u16(v[j][k] ^ 0x80008000u, 0x80008000u, acc);   // sum |x|
    // wave sum with DPP (row_shr 1/2/4/8, row_bcast:15 / :31): lane 63 ends up with the total;
    // __shfl_xor would be six ds_bpermute round trips
    int t = (int)acc;
    t += __builtin_amdgcn_update_dpp(0, t, 0x111, 0xf, 0xf, true);
    t += __builtin_amdgcn_update_dpp(0, t, 0x112, 0xf, 0xf, true);
    t += __builtin_amdgcn_update_dpp(0, t, 0x114, 0xf, 0xf, true);
    t += __builtin_amdgcn_update_dpp(0, t, 0x118, 0xf, 0xf, true);
    t += __builtin_amdgcn_update_dpp(0, t, 0x142, 0xa, 0xf, false);
    t += __builtin_amdgcn_update_dpp(0, t, 0x143, 0xc, 0xf, false);
    if (lane == 63) a.block_amp[(int64_t)s * a.max_blocks + b] = (int32_t)((uint32_t)t >> 11);    // int(sum/2048)
}

__global__ __launch_bounds__(256) void gate_scan_kernel(GateArgs a) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= a.n_streams) return;
    const int32_t len = a.stream_len[s];
    if ((uint32_t)len > (uint32_t)a.max_len) {               // refused capture: out_n_bursts = -1, no burst written
        a.out_n_bursts[s] = -1;
        a.out_open_end[s] = 0;
        if (a.out_slot_len)
            for (int k = 0; k < a.max_bursts; k++) {
                a.out_slot_offset[(int64_t)s * a.max_bursts + k] = 0;
                a.out_slot_len[(int64_t)s * a.max_bursts + k] = 0;
            }
        return;
    }
    const int32_t nb = len / kListenBlock;
    const int32_t* amp = a.block_amp + (int64_t)s * a.max_blocks;
    int32_t* bs = a.out_burst_start + (int64_t)s * a.max_bursts;
    int32_t* bl = a.out_burst_len + (int64_t)s * a.max_bursts;
    int n = 0, open_end = 0;
    int mode = 0;                 // 0 = discard next block, 1 = wait for start, 2 = recording
    int start = 0;
    for (int b = 0; b < nb && n < a.max_bursts; b++) {
        if (mode == 0) { mode = 1; continue; }                       // ref:303
        const int32_t v = amp[b];
        if (mode == 1) {
            if (v > a.amp_start) { start = b; mode = 2; }            // ref:306-309
        } else if (v < a.amp_end) {                                  // ref:316-318 (block included)
            bs[n] = start * kListenBlock;
            bl[n] = (b - start + 1) * kListenBlock;
            n++;
            mode = 0;
        }
    }
    if (mode == 2 && n < a.max_bursts) {     // capture ended while recording: open-ended burst
        bs[n] = start * kListenBlock;
        bl[n] = (nb - start) * kListenBlock;
        n++;
        open_end = 1;
    }
    a.out_n_bursts[s] = n;
    a.out_open_end[s] = open_end;
    // the bursts as demodulator slots (afsk_gate_batch_slots): what Receiver.receive hands to __decodeBits (ref:402-417),
    // ready for afsk_demod_batch* without a host round trip or any arithmetic in between; unused slots get length 0
    // (the demodulator answers them with AFSK_ST_TOO_SHORT and reads nothing)
    if (a.out_slot_len) {
        const int64_t base = a.stream_offset[s];
        for (int k = 0; k < a.max_bursts; k++) {
            a.out_slot_offset[(int64_t)s * a.max_bursts + k] = k < n ? base + bs[k] : 0;
            a.out_slot_len[(int64_t)s * a.max_bursts + k] = k < n ? bl[k] : 0;
        }
    }
}

hipError_t launch_gate(const GateArgs& a, hipStream_t stream) {
    if (a.n_streams <= 0) return hipSuccess;
    if (a.max_blocks > 0) {
        const int64_t waves = (int64_t)a.n_streams * a.max_blocks;
        const int64_t blocks = (waves + 3) / 4;
        if (blocks > 0x7fffffffll) return hipErrorInvalidValue;
        hipLaunchKernelGGL(block_amp_kernel, dim3((uint32_t)blocks), dim3(256), 0, stream, a);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(gate_scan_kernel, dim3((a.n_streams + 255) / 256), dim3(256), 0, stream, a);
    return hipGetLastError();
}

}  // namespace afsk
