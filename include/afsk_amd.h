/*
 * afsk_amd.h -- C-ABI of libafsk_amd.so: the MI355X (gfx950) batched AFSK
 * demodulation path behind lavajuno/afskmodem's Receiver.
 *
 * The reference has no FFI: its hot path is the private method
 *     Receiver.__decodeBits(frames) -> str          (afskmodem.py:354-381)
 * followed by ECC.decode + __bitsToBytes           (afskmodem.py:154-163, 393-399)
 * called from Receiver.load (:420-430) and Receiver.receive (:402-417).
 * The entry points below are what a ctypes binding inside afskmodem.py would
 * bind to replace exactly that span (INTEGRATION.md shows the stub).
 *
 * Conventions
 *  - plain pointers and sizes only; no C++/torch types cross this boundary;
 *  - every function returns 0 on success or a negative AFSK_E_* code and never
 *    throws; afsk_last_error() returns the message of the calling thread's
 *    last failure;
 *  - the caller owns every buffer.  Unless a function name ends in _host, all
 *    data pointers are DEVICE pointers of the current HIP device and the call
 *    is asynchronous on `hip_stream` (a hipStream_t, NULL = default stream);
 *  - there is no CPU fallback: without a HIP device every compute entry fails
 *    with AFSK_E_NO_DEVICE;
 *  - host threads (the I/O pool of the file entries, the staging copies of the
 *    _host entries) are sized by the CPUs the process may really use: the
 *    cgroup CPU quota / affinity mask, divided by LOCAL_WORLD_SIZE when that is
 *    set (one process per GPU: the ranks of a node share the node's CPUs);
 *    AFSK_IO_THREADS / AFSK_COPY_THREADS override.
 */
#ifndef AFSK_AMD_H
#define AFSK_AMD_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AFSK_ABI_VERSION 2 /* 2: afsk_group_plan_* / afsk_demod_batch_grouped, AFSK_ST_BAD_LENGTH, afsk_wav_egress */

/* return codes */
#define AFSK_OK 0
#define AFSK_E_INVALID_ARG (-1)
#define AFSK_E_INVALID_BAUD (-2) /* bit_frames not a multiple of 4 or 2*bf >= 4096      */
#define AFSK_E_NO_DEVICE (-3)
#define AFSK_E_HIP (-4)          /* a HIP runtime call failed, see afsk_last_error      */
#define AFSK_E_HOST (-5)         /* host-side failure in a host-buffer entry (out of memory,
                                    thread creation); nothing is thrown across the boundary */

/* per-stream status written to out_status (reference behaviour in brackets) */
#define AFSK_ST_OK 0
#define AFSK_ST_TOO_SHORT 1 /* len < 4096 [__recoverClockIndex -> -1 -> "" , :323-325] */
#define AFSK_ST_NO_DATA 2   /* no terminator / zero bits [bits == "" -> b"", :422-424]  */
#define AFSK_ST_INVALID_BAUD 3 /* bit_frames[s] rejected (host wrappers raise before launch) */
#define AFSK_ST_BAD_LENGTH 4 /* stream_len[s] < 0 or > AFSK_MAX_STREAM_LEN: refused by the kernel before any
                                sample is addressed (the device entries never see the length array on the
                                host; the host entries reject such a batch with AFSK_E_INVALID_ARG) */

/* Fixed constants of the reference that the kernels compile in. */
#define AFSK_SAMPLE_RATE 48000 /* :69,71,187,233,260,277 */
#define AFSK_SYNC_WINDOW 4096  /* :323,327                */
#define AFSK_DEAD_ZONE 512     /* :290-292                */
#define AFSK_TAIL_SILENCE 4800 /* :468                    */
/* Longest stream the kernels address with 32-bit byte offsets (about 6.2 hours of audio). */
#define AFSK_MAX_STREAM_LEN ((1 << 30) - (1 << 15))

int afsk_version(void);
/* Copies the calling thread's last error message (NUL terminated, truncated to
 * cap) and returns its full length. */
int afsk_last_error(char *buf, int cap);
/* Number of visible HIP devices (0 when there is none); never fails. */
int afsk_device_count(void);
/* hipStreamSynchronize(hip_stream). */
int afsk_sync(void *hip_stream);

/*
 * Replaces Receiver.__decodeBits (:354-381) + ECC.decode (:154-163) +
 * __bitsToBytes (:393-399) for n_streams independent streams at once.
 *
 *  samples        int16 mono 48 kHz, all streams in one allocation
 *  stream_offset  [n] first sample of stream s, in samples from `samples`.  Any value works (streams
 *                 need 2-byte alignment only); EVEN offsets from a dword-aligned `samples` -- best:
 *                 multiples of 8 samples -- run at full speed: a stream that starts on an odd sample
 *                 is fetched by 2-byte-aligned requests, about 25 % slower (same results)
 *  stream_len     [n] length of stream s in samples (0 ... AFSK_MAX_STREAM_LEN; anything else:
 *                 status AFSK_ST_BAD_LENGTH for that stream, its neighbours are unaffected)
 *  bit_frames     [n] 48000 / baud of stream s (Receiver.__init__ :277);
 *                 must be a multiple of 4 with 2*bit_frames < 4096
 *  amp_end_threshold  Receiver(amp_end_threshold=...) (:276), squelch of :375
 *  out_bytes      [n, out_stride] decoded payload bytes (row s, first
 *                 min(out_nbytes[s], out_stride) bytes are written; any
 *                 stride works -- rows of more than ~200 bytes are cheapest
 *                 when out_bytes and out_stride are multiples of 128: what a
 *                 launch pays for its output is the number of cache lines
 *                 it dirties)
 *  out_nbytes     [n] number of decoded bytes (may exceed out_stride: the row
 *                 was then truncated; size rows as stream_len/(14*bf)+1)
 *  out_nbits      [n] coded bits demodulated incl. ECC (debug line :380)
 *  out_clock_idx  [n] recovered clock index (:338), -1 when too short
 *  out_term_frame [n] frame index after the training terminator (:368), -1
 *                 when too short
 *  out_status     [n] AFSK_ST_*
 *
 * Results are bit-exact with the reference for every input (integer path).
 */
int afsk_demod_batch(const int16_t *samples, const int64_t *stream_offset,
                     const int32_t *stream_len, const int32_t *bit_frames,
                     int32_t amp_end_threshold, int32_t n_streams, uint8_t *out_bytes,
                     int32_t out_stride, int32_t *out_nbytes, int32_t *out_nbits,
                     int32_t *out_clock_idx, int32_t *out_term_frame, int32_t *out_status,
                     void *hip_stream);

/*
 * afsk_demod_batch plus two optional soft outputs (either may be NULL), for callers that
 * want link-quality figures without a second pass over the samples.  Both are values the
 * reference computes and discards:
 *
 *  out_corrected  [n] codewords of ECC.decode's input (floor(nbits/7) of them, :156-157) whose
 *                 syndrome (:146-147) was non-zero, i.e. single-bit corrections applied
 *  out_margins    [n, margin_stride] per demodulated symbol k (sample clock_idx + k*bf),
 *                 space_diff - mark_diff of __decodeBit (:348-349): > 0 decodes as 1, <= 0
 *                 as 0 (:350-351).  Written for the symbols the reference demodulated:
 *                 k < (term_frame - clock_idx)/bf + nbits when a terminator was found, else
 *                 every k with clock_idx + k*bf < len - bf; entries past that (and past
 *                 margin_stride) are unspecified/not written.
 */
int afsk_demod_batch_ex(const int16_t *samples, const int64_t *stream_offset,
                        const int32_t *stream_len, const int32_t *bit_frames,
                        int32_t amp_end_threshold, int32_t n_streams, uint8_t *out_bytes,
                        int32_t out_stride, int32_t *out_nbytes, int32_t *out_nbits,
                        int32_t *out_clock_idx, int32_t *out_term_frame, int32_t *out_status,
                        int32_t *out_corrected, int32_t *out_margins, int32_t margin_stride,
                        void *hip_stream);

/*
 * The Receiver-shaped form of afsk_demod_batch_ex: ONE bit_frames for every stream of the
 * launch, passed by value.  A Receiver has exactly one baud rate (Receiver.__init__ :275-284:
 * bit_frames, the two tone templates and the training cycle are per-object constants), so a batch
 * decoded on behalf of one Receiver never needs the per-stream array.  bit_frames is validated on
 * the host (AFSK_E_INVALID_BAUD) and selects a kernel compiled for exactly that geometry -- no
 * per-stream load of it, no geometry switch in the kernel.  Same outputs, bit for bit, as
 * afsk_demod_batch_ex with bit_frames[s] == bit_frames for all s; out_corrected / out_margins may
 * be NULL (margin_stride 0).
 */
int afsk_demod_batch_uniform(const int16_t *samples, const int64_t *stream_offset,
                             const int32_t *stream_len, int32_t bit_frames,
                             int32_t amp_end_threshold, int32_t n_streams, uint8_t *out_bytes,
                             int32_t out_stride, int32_t *out_nbytes, int32_t *out_nbits,
                             int32_t *out_clock_idx, int32_t *out_term_frame, int32_t *out_status,
                             int32_t *out_corrected, int32_t *out_margins, int32_t margin_stride,
                             void *hip_stream);

/*
 * Rate-grouped dispatch of a MIXED-baud batch whose bit_frames the HOST can see (a list of Receivers of
 * different baud rates, each with its own streams: one baud rate per Receiver, :275-284; streams are
 * independent, :354-381).  The plan buckets the streams by bit_frames once (a stable sort on the host, one
 * upload of an index list + the bit_frames, 8 bytes per stream); afsk_demod_batch_grouped then decodes the
 * batch with ONE kernel launch that walks the streams bucket by bucket (inside windows of 4096 consecutive streams,
 * so that the streams in flight stay close in memory), so that the wavefronts resident on a
 * compute unit run the same rate's code: 3 - 12 % faster than stream order when four or more rates are mixed
 * (below that the plan keeps stream order; one rate: the kernel of afsk_demod_batch_uniform).  Nothing but a
 * kernel launch: asynchronous on hip_stream like every device entry, safe inside a stream capture, and
 * calls may share a plan freely.  Outputs land at the ORIGINAL stream numbers, bit for bit what
 * afsk_demod_batch_ex writes for the same bit_frames[] (a stream with an invalid bit_frames gets status
 * AFSK_ST_INVALID_BAUD).  afsk_demod_batch (bit_frames[] in device memory, streams in the caller's order)
 * stays the entry for rates only the device knows.
 * (One launch of the uniform kernel per rate on forked HIP streams was measured and rejected: DESIGN.md 4.4.)
 *
 *  afsk_group_plan_create   bit_frames_host: HOST array [n].  Allocates 8 n bytes on the current device and
 *                           fills them (synchronous).
 *  afsk_group_plan_create_ragged  (r6, an addition: ABI version unchanged) the same with stream_len_host, a HOST
 *                           array [n] of the lengths the launches will be given (NULL = afsk_group_plan_create).  One
 *                           wavefront decodes one stream whatever its length and a workgroup of four keeps its share of
 *                           a CU until its longest stream ends, so when the lengths differ (shortest below 3/4 of the
 *                           longest) the walk takes, inside every window of 4096 consecutive streams and every rate,
 *                           the LONGEST streams first -- also for a one-rate batch, whose uniform kernel then walks the
 *                           list.  Results do not depend on it (same outputs at the same stream numbers); lengths that
 *                           differ from those given later only cost speed.  The host entries do this by themselves.
 *  afsk_group_plan_info     n_streams, number of buckets, and per bucket (first `cap` of them, in launch
 *                           order: largest first) its bit_frames (0 = the refused streams) and stream count;
 *                           any pointer may be NULL
 *  afsk_demod_batch_grouped every array as afsk_demod_batch_ex, indexed by the original stream number;
 *                           n_streams is the plan's
 *  afsk_group_plan_destroy  after the launches that use the plan have completed (NULL is fine)
 */
typedef struct afsk_group_plan afsk_group_plan;
int afsk_group_plan_create(const int32_t *bit_frames_host, int32_t n_streams, afsk_group_plan **out_plan);
int afsk_group_plan_create_ragged(const int32_t *bit_frames_host, const int32_t *stream_len_host, int32_t n_streams,
                                  afsk_group_plan **out_plan);
int afsk_group_plan_info(const afsk_group_plan *plan, int32_t *out_n_streams, int32_t *out_n_groups,
                         int32_t *out_group_bit_frames, int32_t *out_group_count, int32_t cap);
int afsk_group_plan_destroy(afsk_group_plan *plan);
int afsk_demod_batch_grouped(const afsk_group_plan *plan, const int16_t *samples,
                             const int64_t *stream_offset, const int32_t *stream_len,
                             int32_t amp_end_threshold, uint8_t *out_bytes, int32_t out_stride,
                             int32_t *out_nbytes, int32_t *out_nbits, int32_t *out_clock_idx,
                             int32_t *out_term_frame, int32_t *out_status, int32_t *out_corrected,
                             int32_t *out_margins, int32_t margin_stride, void *hip_stream);

/*
 * Same operation on HOST buffers: allocates device scratch, copies in, runs the
 * HIP kernel, copies out, synchronises.  This is the PCIe-inclusive convenience
 * path a single Receiver.load() uses; it is not the benchmarked entry.
 * Both host entries work on a private NON-BLOCKING HIP stream of the calling thread (never the
 * NULL stream): they do not synchronise with the caller's own streams or with calls made by
 * other threads, and may be called concurrently (the reference's Receivers are independent
 * objects, afskmodem.py:275-284).  The host can see bit_frames[] here: one value -> the uniform kernel of
 * afsk_demod_batch_uniform, several -> the rate-sorted launch of afsk_demod_batch_grouped.  stream_len[s] above AFSK_MAX_STREAM_LEN is rejected here (AFSK_E_INVALID_ARG).
 */
int afsk_demod_batch_host(const int16_t *samples, int64_t total_samples,
                          const int64_t *stream_offset, const int32_t *stream_len,
                          const int32_t *bit_frames, int32_t amp_end_threshold,
                          int32_t n_streams, uint8_t *out_bytes, int32_t out_stride,
                          int32_t *out_nbytes, int32_t *out_nbits, int32_t *out_clock_idx,
                          int32_t *out_term_frame, int32_t *out_status);

/*
 * Gather form of the host entry: stream s is the host array streams[s] of stream_len[s]
 * samples (no concatenated buffer needed -- e.g. one array per .wav file or per
 * Receiver.load()-style frame list).  The streams are packed through two pinned 32 MiB
 * windows by a few copy threads while the previous window is on the PCIe link, then
 * demodulated with one launch.  Calls serialise on the cached staging buffers.
 */
int afsk_demod_streams_host(const int16_t *const *streams, const int32_t *stream_len,
                            const int32_t *bit_frames, int32_t amp_end_threshold,
                            int32_t n_streams, uint8_t *out_bytes, int32_t out_stride,
                            int32_t *out_nbytes, int32_t *out_nbits, int32_t *out_clock_idx,
                            int32_t *out_term_frame, int32_t *out_status);

/* The host entries keep their device scratch (samples + results) allocated between
 * calls (and the gather entry its pinned windows), because hipMalloc of a large buffer costs
 * far more than the transfer; this frees them.
 * Safe to call at any time; the next host-entry call allocates again. */
int afsk_host_scratch_release(void);

/*
 * .wav container ingest at scale (replaces SoundInput.loadFromFile,
 * afskmodem.py:213-217, for many files).  Two steps so that the caller owns the device buffer:
 *
 * afsk_wav_probe   walks the RIFF chunks of every file exactly like the stdlib `wave` reader the
 *   reference calls (:214): 'RIFF' <size> 'WAVE', then chunks with even padding; 'fmt ' must
 *   come before 'data'; the walk stops at 'data'.  Like the reference, rate / width / channel
 *   count are NOT interpreted -- they only bound the byte count:
 *   readframes(getnframes()) returns (data_size / (channels * bytes_per_sample)) whole frames,
 *   clipped to what the file really holds.  Host-only (no HIP call); parallel over files.
 *     out_data_offset [n] byte offset of the data chunk's payload in the file
 *     out_data_bytes  [n] bytes readframes(getnframes()) would return (may be odd:
 *                         __convertFrames (:201-205) then drops the last byte)
 *     out_status      [n] AFSK_WAV_*; anything but AFSK_WAV_OK means "not a plain PCM RIFF file,
 *                         or unreadable" -- the Python host re-opens such a file with the stdlib
 *                         reader so that the caller sees the reference's own exception
 *
 * afsk_wav_upload  reads data_bytes[s] & ~1 bytes at data_offset[s] of file s with pread()
 *   STRAIGHT INTO the library's two pinned staging windows (one copy: page cache -> pinned
 *   memory, a few threads in parallel) and sends each window to
 *   d_samples[stream_offset[s] ..] while the next one is being filled.  stream_offset is in
 *   samples, ascending, streams must not overlap and must fit capacity_samples.  Synchronous for
 *   the caller, on the calling thread's private non-blocking stream (NOT ordered against the
 *   caller's own streams: work of the caller that still reads or writes the destination range
 *   must have completed before the call).  Device bytes written: the streams themselves, plus
 *   gaps of at most 256 bytes BETWEEN two consecutive streams (alignment padding), which are set
 *   to zero; a larger gap between two streams and everything outside the streams is not touched,
 *   so one buffer can be filled by several calls.  afsk_wav_probe is host-only and, like
 *   afsk_wav_upload's file reading, fork-safe (a forked child builds its own I/O thread pool).
 */
#define AFSK_WAV_OK 0
#define AFSK_WAV_IO 1          /* cannot open / read                                      */
#define AFSK_WAV_NOT_RIFF 2    /* no 'RIFF' .. 'WAVE' header                              */
#define AFSK_WAV_NO_DATA 3     /* 'fmt ' and/or 'data' chunk missing, or 'data' first     */
#define AFSK_WAV_FORMAT 4      /* format tag other than PCM, zero channels / sample width */
#define AFSK_WAV_SLOT 5        /* afsk_wav_ingest only: the data does not fit the caller's slot    */
int afsk_wav_probe(const char *const *paths, int32_t n_files, int64_t *out_data_offset,
                   int64_t *out_data_bytes, int32_t *out_status);
int afsk_wav_upload(const char *const *paths, const int64_t *data_offset, const int64_t *data_bytes,
                    const int64_t *stream_offset, int32_t n_files, int16_t *d_samples,
                    int64_t capacity_samples);

/*
 * The same ingest in ONE pass per file (what batch.load_wav_batch uses since r3):
 *
 * afsk_file_sizes  st_size of every file (-1: cannot stat), host-only, parallel.  A file's data chunk
 *   cannot be longer than the file, so the sizes give a device layout BEFORE any file is opened:
 *   slot i = the device range reserved for file i (slot_offset[i], slot_samples[i], in samples).
 * afsk_wav_ingest  per file: open, the chunk walk of afsk_wav_probe, pread of the data chunk straight
 *   into pinned memory, close -- pipelined against the H2D copies (four 16 MiB staging buffers; pool
 *   threads fill, the calling thread sends).  out_data_offset / out_data_bytes / out_status as
 *   afsk_wav_probe; the stream of file i is d_samples[slot_offset[i] .. + out_data_bytes[i] / 2).
 *   Device bytes written: every slot in full (the data, then zeros) and gaps of at most 256 bytes
 *   between consecutive slots (zeros); a file whose status is not AFSK_WAV_OK -- or whose data would not
 *   fit its slot: AFSK_WAV_SLOT -- leaves a zeroed slot for the caller to fill (batch.load_wav_batch
 *   re-opens it with the stdlib reader, so the caller sees the reference's own exception or data).
 *   Slots ascending, non-overlapping, inside capacity_samples.  Synchronous for the caller, on the
 *   calling thread's private non-blocking stream (same ordering contract as afsk_wav_upload); fork-safe.
 */
int afsk_file_sizes(const char *const *paths, int32_t n_files, int64_t *out_size_bytes);
int afsk_wav_ingest(const char *const *paths, int32_t n_files, const int64_t *slot_offset,
                    const int64_t *slot_samples, int16_t *d_samples, int64_t capacity_samples,
                    int64_t *out_data_offset, int64_t *out_data_bytes, int32_t *out_status);

/*
 * .wav EGRESS at scale (r4; the mirror image of afsk_wav_ingest, for Transmitter.save, afskmodem.py:481-484, of many
 * payloads): stream s of the device buffer, d_samples[stream_offset[s] .. + stream_len[s]), is written to paths[s] as the
 * file SoundOutput.writeToFile (:256-263) produces through the stdlib writer -- the canonical 44-byte RIFF/WAVE header
 * (PCM, 1 channel, 48000 Hz, 16 bit, data size 2 * stream_len[s]) followed by the samples.  (The decimate / duplicate
 * quirk of SoundOutput.__convertFrames, :239-244, is the MODULATOR's job: afsk_modulate_batch(wav_quirk = 1).)
 * stream_offset (samples, ascending, non-overlapping) and stream_len are HOST arrays.  The calling thread copies windows
 * of the device range into the pinned staging ring (D2H, two alternating streams) and pool threads write each window's
 * file pieces as soon as its copy has completed: one open / pwritev (header + data) / close per file that lies inside
 * one window.  Existing files are replaced.  out_status[s]: AFSK_WAV_OK or AFSK_WAV_IO (cannot create / write: that
 * file's problem, the batch goes on).  Synchronous for the caller, on the calling thread's private streams (work of the
 * caller that still writes the source range must have completed before the call); fork-safe like the ingest.
 */
int afsk_wav_egress(const char *const *paths, int32_t n_files, const int16_t *d_samples,
                    const int64_t *stream_offset, const int32_t *stream_len, int32_t *out_status);

/*
 * On-device input synthesis: Transmitter.__getFrames (:452-469) with ECC.encode
 * (:166-175) and, when wav_quirk != 0, SoundOutput.__convertFrames' decimate-by-2
 * + duplicate (:239-244), written to samples[stream_offset[s] .. +stream_len[s])
 * (truncated, or zero padded after the 4800-sample tail).
 *
 *  payload      uint8 [n, payload_stride]; payload_len [n] bytes used per row
 *  bit_frames   [n] 48000 / baud; a positive multiple of 4 (every baud rate the reference's
 *               Waveforms accept, :69-70/:81-82, gives one); any other value yields an
 *               all-zero stream (device arrays are not validated on the host)
 *  ts_cycles    [n] int(baud * training_time / 2)   (:438)
 *  max_stream_len  host-side upper bound of stream_len[] (sizes the grid); a stream whose device-side
 *               stream_len[s] is negative or above it is skipped (nothing written for it)
 */
int afsk_modulate_batch(const uint8_t *payload, int32_t payload_stride,
                        const int32_t *payload_len, const int32_t *bit_frames,
                        const int32_t *ts_cycles, const int64_t *stream_offset,
                        const int32_t *stream_len, int32_t max_stream_len, int32_t n_streams,
                        int32_t wav_quirk, int16_t *samples, void *hip_stream);

/*
 * Batched replay of Receiver.__listen (:299-319) over finite captures: the live-input gate
 * that decides which 2048-frame blocks (:189, :209) of a continuous recording are handed to
 * __decodeBits.  Per capture, repeated receive() calls are emulated: each call discards one
 * block (:303), waits for a block with getAmplitude > amp_start_threshold (:306) and records
 * blocks up to and including the first one with getAmplitude < amp_end_threshold (:316).
 * Only whole blocks of the capture are considered (no timeout: the call scans to the end).
 *
 *  max_stream_len   host-side upper bound of stream_len[]; max_blocks = max_stream_len / 2048.  A capture
 *                   whose device-side stream_len[s] is negative or above it is refused:
 *                   out_n_bursts[s] = -1, no sample of it is read
 *  block_amp        workspace AND output, int32 [n, max_blocks]: int(sum|x| / 2048) per block
 *  out_n_bursts     [n] bursts found (<= max_bursts)
 *  out_burst_start  [n, max_bursts] first sample of burst k, relative to the stream start
 *  out_burst_len    [n, max_bursts] samples in burst k (a multiple of 2048)
 *  out_open_end     [n] 1 when the last burst reached the end of the capture without a quiet
 *                   block (a live receiver would still be recording)
 * The bursts are demodulated by passing stream_offset[s] + out_burst_start[s,k] and
 * out_burst_len[s,k] to afsk_demod_batch.
 *
 * afsk_gate_batch_slots (r6, an addition) does that arithmetic on the device as well: it also writes the bursts as
 * FIXED demodulator slots -- slot s * max_bursts + k = burst k of capture s:
 *  out_slot_offset  int64 [n, max_bursts] stream_offset[s] + out_burst_start[s,k]; 0 where capture s has fewer bursts
 *  out_slot_len     int32 [n, max_bursts] out_burst_len[s,k]; 0 where capture s has fewer bursts (afsk_demod_batch*
 *                   answers such a slot with AFSK_ST_TOO_SHORT and reads nothing)
 * so that  afsk_gate_batch_slots(...) ; afsk_demod_batch_uniform(samples, out_slot_offset, out_slot_len, bit_frames,
 * amp_end, n * max_bursts, ...)  on one stream is what repeated Receiver.receive() calls do (:402-417) for n captures,
 * with no host round trip in between (two launches + one: capturable into one HIP graph).
 */
int afsk_gate_batch(const int16_t *samples, const int64_t *stream_offset,
                    const int32_t *stream_len, int32_t max_stream_len,
                    int32_t amp_start_threshold, int32_t amp_end_threshold, int32_t n_streams,
                    int32_t max_bursts, int32_t *block_amp, int32_t *out_n_bursts,
                    int32_t *out_burst_start, int32_t *out_burst_len, int32_t *out_open_end,
                    void *hip_stream);
int afsk_gate_batch_slots(const int16_t *samples, const int64_t *stream_offset,
                          const int32_t *stream_len, int32_t max_stream_len,
                          int32_t amp_start_threshold, int32_t amp_end_threshold, int32_t n_streams,
                          int32_t max_bursts, int32_t *block_amp, int32_t *out_n_bursts,
                          int32_t *out_burst_start, int32_t *out_burst_len, int32_t *out_open_end,
                          int64_t *out_slot_offset, int32_t *out_slot_len, void *hip_stream);

/*
 * Deterministic additive noise (build-owned test/benchmark input generator, no
 * reference counterpart): per sample an Irwin-Hall sum of 16 uniform u16 drawn
 * from a counter hash of (seed, stream_idx_base + s, sample index), centred,
 * multiplied by scale_q24[s] / 2^24, added and clipped to int16.  Integer-only,
 * so the CPU oracle's generator produces identical samples.
 */
int afsk_add_noise_batch(int16_t *samples, const int64_t *stream_offset,
                         const int32_t *stream_len, int32_t max_stream_len,
                         const int32_t *scale_q24, int32_t n_streams, uint32_t seed,
                         uint32_t stream_idx_base, void *hip_stream);

#ifdef __cplusplus
}
#endif
#endif
