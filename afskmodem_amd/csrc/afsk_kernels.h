// Internal header shared by the HIP translation units of libafsk_amd.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace afsk {

// Which block of work a workgroup takes (r5; demod kernels: 4 streams, block_amp_kernel: 4 listen blocks).  Workgroups are dealt round-robin over the 8 XCDs (observed, never a
// contract: a different placement only costs the speed), and every XCD has its own L2 -- with blockIdx.x * WPB the
// 32 streams whose status words share one 128-byte line of out_nbytes[] ... (and whose output rows are neighbours) are
// written by 8 blocks on 8 different XCDs: every 4- or 54-byte store leaves its L2 as a partial line of its own, and
// small writes threaded into a streaming read cost the memory system far more than their bytes (4 - 10 % of the
// kernel for 0.1 - 0.5 % of the traffic: profiles/r5_exp24_store_forms.txt).  Inside every group of 64 consecutive
// blocks the 8 blocks of one XCD (equal blockIdx % 8, dispatched together) take 8 CONSECUTIVE blocks of streams -- one
// line of every status array, 32 neighbouring rows -- so their stores meet in one L2 and leave it merged.  The launch
// as a whole still walks the input front to back (handing every XCD one contiguous eighth of the streams was 1 - 6 %
// SLOWER: profiles/r5_exp25_xcd_remap.txt).  A permutation of [0, nwg): identity in the last, partial group.
__device__ __forceinline__ int xcd_block(int bid, int nwg) {
#ifdef AFSK_NO_XCD_REMAP
    return bid;
#else
    if (bid >= (nwg & ~63)) return bid;
    return (bid & ~63) | ((bid & 7) << 3) | ((bid >> 3) & 7);
#endif
}

struct DemodArgs {
    const int16_t* samples;
    const int64_t* stream_offset;
    const int32_t* stream_len;
    const int32_t* bit_frames;                    // [n] per-stream values (mixed-baud kernel); unused by the uniform kernels
    int32_t amp_end;
    int32_t n_streams;
    uint8_t* out_bytes;
    int32_t out_stride;
    int32_t* out_nbytes;
    int32_t* out_nbits;
    int32_t* out_clock_idx;
    int32_t* out_term_frame;
    int32_t* out_status;
    unsigned long long* debug_stamps = nullptr;   // diagnostics only (tools/kbench): 4 x u64 per stream
    // optional soft outputs (afsk_demod_batch_ex); null = not wanted
    int32_t* out_corrected = nullptr;             // [n] codewords with a non-zero Hamming syndrome
    int32_t* out_margins = nullptr;               // [n, margin_stride] space_diff - mark_diff per symbol
    int32_t margin_stride = 0;
    int32_t uniform_bit_frames = 0;               // the one bit_frames of a uniform launch (afsk_demod_batch_uniform)
    // grouped dispatch (afsk_demod_batch_grouped): wave w of a per-stream launch decodes stream stream_index[w]
    // (the rate-sorted list of all n_streams); every per-stream array, inputs and outputs, is addressed by that
    // stream number.  null = wave w decodes stream w.
    const int32_t* stream_index = nullptr;
};

// Longest stream the kernels address (32-bit byte offsets into a stream): AFSK_MAX_STREAM_LEN of the C-ABI.
constexpr int32_t kMaxStreamLen = (1 << 30) - (1 << 15);

struct ModulateArgs {
    const uint8_t* payload;
    int32_t payload_stride;
    const int32_t* payload_len;
    const int32_t* bit_frames;
    const int32_t* ts_cycles;
    const int64_t* stream_offset;
    const int32_t* stream_len;
    int32_t n_streams;
    int32_t wav_quirk;
    int16_t* samples;
    int32_t chunks;   // blocks per stream (set by the launcher)
    int32_t max_len;  // the caller's bound of stream_len[] (set by the launcher): longer / negative entries are skipped
};

struct NoiseArgs {
    int16_t* samples;
    const int64_t* stream_offset;
    const int32_t* stream_len;
    const int32_t* scale_q24;
    int32_t n_streams;
    uint32_t seed;
    uint32_t stream_idx_base;
    int32_t chunks;
    int32_t max_len;  // as ModulateArgs::max_len
};

struct GateArgs {
    const int16_t* samples;
    const int64_t* stream_offset;
    const int32_t* stream_len;
    int32_t amp_start;
    int32_t amp_end;
    int32_t n_streams;
    int32_t max_len;        // the caller's bound of stream_len[]: longer / negative entries are refused
    int32_t max_blocks;     // workspace row length = max_stream_len / 2048
    int32_t max_bursts;
    int32_t* block_amp;     // [n_streams, max_blocks]
    int32_t* out_n_bursts;
    int32_t* out_burst_start;
    int32_t* out_burst_len;
    int32_t* out_open_end;
    // optional (afsk_gate_batch_slots): the bursts as fixed demodulator slots -- slot s * max_bursts + k = burst k of
    // capture s as (absolute first sample, length); length 0 (and offset 0) where capture s has fewer bursts
    int64_t* out_slot_offset = nullptr;   // [n_streams, max_bursts]
    int32_t* out_slot_len = nullptr;      // [n_streams, max_bursts]
};

hipError_t launch_gate(const GateArgs& a, hipStream_t stream);
hipError_t launch_demod(const DemodArgs& a, hipStream_t stream);
// one bit_frames (a.uniform_bit_frames, host-validated) for every stream of the launch
hipError_t launch_demod_uniform(const DemodArgs& a, hipStream_t stream);
hipError_t launch_modulate(ModulateArgs a, int32_t max_len, hipStream_t stream);
hipError_t launch_noise(NoiseArgs a, int32_t max_len, hipStream_t stream);

}  // namespace afsk
