/*
 * afsk_oracle.c -- TEST INFRASTRUCTURE ONLY (see afsk_oracle.h).
 *
 * A deliberately literal, scalar restatement of the reference algorithm: every
 * SAD is summed sample by sample, every mean is a double division truncated
 * toward zero exactly like Python's int(total / n), the sync search is the
 * brute-force sweep.  None of the prefix-sum / threshold tricks the HIP kernel
 * uses appear here, so agreement between the two is an independent check.
 *
 * "ref:" = /root/reference/afskmodem.py line numbers.
 */
#include "afsk_oracle.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#define SAMPLE_RATE 48000
#define SYNC_WINDOW 4096   /* ref:323, 327 */
#define DEAD_ZONE 512      /* ref:290-292 */
#define TAIL_SILENCE 4800  /* ref:468 */
#define HI 32767
#define LO (-32768)

/* ---------------------------------------------------------------- Waveforms */

/* ref:68-77.  bit_frames is the float 48000/baud; each half is int(bit_frames/2). */
int afsk_o_space_tone(int baud, int16_t *out, int cap) {
    if (baud <= 0 || SAMPLE_RATE % baud != 0) return AFSK_O_ERR_INVALID_BAUD; /* ref:69-70 */
    double bit_frames = (double)SAMPLE_RATE / (double)baud;                   /* ref:71 */
    int half = (int)(bit_frames / 2.0);                                       /* ref:73,75 */
    if (2 * half > cap) return AFSK_O_ERR_CAPACITY;
    for (int i = 0; i < half; i++) out[i] = HI;                               /* ref:73-74 */
    for (int i = 0; i < half; i++) out[half + i] = LO;                        /* ref:75-76 */
    return 2 * half;
}

/* ref:80-85: the mark tone is two space tones of twice the baud rate. */
int afsk_o_mark_tone(int baud, int16_t *out, int cap) {
    if (baud <= 0 || SAMPLE_RATE % baud != 0) return AFSK_O_ERR_INVALID_BAUD; /* ref:81-82 */
    int n1 = afsk_o_space_tone(baud * 2, out, cap);                           /* ref:83 */
    if (n1 < 0) return n1;
    int n2 = afsk_o_space_tone(baud * 2, out + n1, cap - n1);                 /* ref:84 */
    if (n2 < 0) return n2;
    return n1 + n2;
}

/* ref:88-91: mark followed by space. */
int afsk_o_training_cycle(int baud, int16_t *out, int cap) {
    int n1 = afsk_o_mark_tone(baud, out, cap);                                /* ref:89 */
    if (n1 < 0) return n1;
    int n2 = afsk_o_space_tone(baud, out + n1, cap - n1);                     /* ref:90 */
    if (n2 < 0) return n2;
    return n1 + n2;
}

/* ref:94-98.  abs(-32768) is 32768 in Python, so widen before abs. */
int afsk_o_get_amplitude(const int16_t *frames, int n) {
    int64_t sum = 0;
    for (int i = 0; i < n; i++) {
        int32_t v = frames[i];
        sum += v < 0 ? -v : v;
    }
    return (int)((double)sum / (double)n); /* int(sum / len) truncates */
}

/* ref:101-107 (length check is the caller's: the C signature has one n). */
int afsk_o_get_diff(const int16_t *a, const int16_t *b, int n) {
    int64_t total = 0;
    for (int i = 0; i < n; i++) {
        int32_t d = (int32_t)a[i] - (int32_t)b[i];
        total += d < 0 ? -d : d;
    }
    return (int)((double)total / (double)n);
}

/* ----------------------------------------------------------------- Receiver */

/* ref:287-296 hard limiter with a +-512 dead zone. */
void afsk_o_amplify(const int16_t *chunk, int n, int16_t *out) {
    for (int i = 0; i < n; i++) {
        if (chunk[i] > DEAD_ZONE) out[i] = HI;
        else if (chunk[i] < -DEAD_ZONE) out[i] = LO;
        else out[i] = 0;
    }
}

/* ref:322-339.  Returns the clock index, -1 when len < 4096, or an error code
 * below -1 (shifted so that -1 keeps the reference's meaning). */
int afsk_o_recover_clock_index(const int16_t *frames, int64_t len, int bit_frames,
                               const int16_t *training_cycle, int training_len) {
    if (len < SYNC_WINDOW) return -1;                        /* ref:323-325 */
    int n_scan = SYNC_WINDOW - bit_frames * 2;               /* ref:327 */
    if (n_scan > 0 && training_len != bit_frames * 2)        /* ref:102-103 via :329 */
        return AFSK_O_ERR_LEN_MISMATCH - 8;
    if (n_scan <= 0) return AFSK_O_ERR_EMPTY_SCAN - 8;       /* ref:332 IndexError */
    int min_diff = 0, min_index = 0;
    for (int i = 0; i < n_scan; i++) {
        int d = afsk_o_get_diff(training_cycle, frames + i, training_len); /* ref:328-331 */
        if (i == 0 || d < min_diff) {                        /* ref:332-337 strict <, first min */
            min_diff = d;
            min_index = i;
        }
    }
    return min_index;
}

/* ref:342-351.  Returns 1 for "1", 0 for "0". */
int afsk_o_decode_bit(const int16_t *chunk, int bit_frames, const int16_t *mark_tone,
                      const int16_t *space_tone) {
    int16_t amp[SAMPLE_RATE];
    afsk_o_amplify(chunk, bit_frames, amp);                         /* ref:344 */
    int mark_diff = afsk_o_get_diff(mark_tone, amp, bit_frames);    /* ref:346 */
    int space_diff = afsk_o_get_diff(space_tone, amp, bit_frames);  /* ref:347 */
    return mark_diff < space_diff ? 1 : 0;                          /* ref:348-351 */
}

/* ref:342-351 again, also reporting space_diff - mark_diff (values the reference computes at
 * :346-347 and discards); the decision is unchanged: 1 iff mark_diff < space_diff. */
static int decode_bit_soft(const int16_t *chunk, int bit_frames, const int16_t *mark_tone,
                           const int16_t *space_tone, int32_t *margin) {
    int16_t amp[SAMPLE_RATE];
    afsk_o_amplify(chunk, bit_frames, amp);
    int mark_diff = afsk_o_get_diff(mark_tone, amp, bit_frames);
    int space_diff = afsk_o_get_diff(space_tone, amp, bit_frames);
    if (margin) *margin = space_diff - mark_diff;
    return mark_diff < space_diff ? 1 : 0;
}

/* ref:386-390 sliding 4-slot window, true when it reads 1,0,0,0. */
static int scan_training(int seq[4], int current) {
    for (int i = 1; i < 4; i++) seq[i - 1] = seq[i];
    seq[3] = current;
    return seq[0] == 1 && seq[1] == 0 && seq[2] == 0 && seq[3] == 0;
}

struct templates {
    int16_t *space, *mark, *training;
    int space_len, mark_len, training_len;
};

static int build_templates_baud(int baud, struct templates *t) {
    int cap = 4 * SAMPLE_RATE;
    t->space = (int16_t *)malloc(sizeof(int16_t) * cap * 3);
    if (!t->space) return AFSK_O_ERR_CAPACITY;
    t->mark = t->space + cap;
    t->training = t->mark + cap;
    t->space_len = afsk_o_space_tone(baud, t->space, cap);       /* ref:280 */
    if (t->space_len < 0) { free(t->space); return t->space_len; }
    t->mark_len = afsk_o_mark_tone(baud, t->mark, cap);          /* ref:281 */
    if (t->mark_len < 0) { free(t->space); return t->mark_len; }
    t->training_len = afsk_o_training_cycle(baud, t->training, cap); /* ref:282 */
    if (t->training_len < 0) { free(t->space); return t->training_len; }
    return AFSK_O_OK;
}

/* Shared body of ref:354-381 once the templates exist. */
static int64_t decode_bits_tpl(const int16_t *frames, int64_t len, int bit_frames,
                               const struct templates *t, int amp_end_threshold,
                               uint8_t *bits_out, int64_t bits_cap, int32_t *clock_idx,
                               int64_t *term_frame, int32_t *margins, int64_t margin_cap) {
    int ci = afsk_o_recover_clock_index(frames, len, bit_frames, t->training, t->training_len);
    if (clock_idx) *clock_idx = ci < -1 ? -1 : ci;
    if (term_frame) *term_frame = -1;
    if (ci < -1) return ci + 8;                                   /* propagated error */
    if (ci == -1) return 0;                                       /* ref:357-358 */
    if (t->mark_len != bit_frames || t->space_len != bit_frames)
        return AFSK_O_ERR_LEN_MISMATCH;                           /* ref:102-103 via :346 */
    int64_t i = ci;
    int training_bits[4] = {0, 0, 0, 0};                          /* ref:361 */
    while (i < len - bit_frames) {                                /* ref:362 */
        const int16_t *chunk = frames + i;                        /* ref:363 */
        int64_t k = (i - ci) / bit_frames;                        /* symbol number (soft output) */
        int32_t m;
        i += bit_frames;                                          /* ref:364 */
        int b = decode_bit_soft(chunk, bit_frames, t->mark, t->space, &m);
        if (margins && k < margin_cap) margins[k] = m;
        if (scan_training(training_bits, b))                      /* ref:365 */
            break;
    }
    if (term_frame) *term_frame = i;                              /* ref:368 */
    int64_t nbits = 0;
    while (i < len - bit_frames) {                                /* ref:372 */
        const int16_t *chunk = frames + i;                        /* ref:373 */
        if (afsk_o_get_amplitude(chunk, bit_frames) < amp_end_threshold) break; /* ref:375-376 */
        if (nbits >= bits_cap) return AFSK_O_ERR_CAPACITY;
        int64_t k = (i - ci) / bit_frames;
        int32_t m;
        bits_out[nbits++] = (uint8_t)decode_bit_soft(chunk, bit_frames, t->mark, t->space, &m);
        if (margins && k < margin_cap) margins[k] = m;
        i += bit_frames;                                          /* ref:377-378 */
    }
    return nbits;
}

int64_t afsk_o_decode_bits(const int16_t *frames, int64_t len, int baud, int amp_end_threshold,
                           uint8_t *bits_out, int64_t bits_cap, int32_t *clock_idx,
                           int64_t *term_frame) {
    struct templates t;
    int rc = build_templates_baud(baud, &t);
    if (rc < 0) return rc;
    int bit_frames = (int)((double)SAMPLE_RATE / (double)baud);   /* ref:277 */
    int64_t n = decode_bits_tpl(frames, len, bit_frames, &t, amp_end_threshold, bits_out,
                                bits_cap, clock_idx, term_frame, NULL, 0);
    free(t.space);
    return n;
}

/* ---------------------------------------------------------------------- ECC */

/* ref:115-123 generator rows, ref:125-129 parity-check rows. */
static const int M_GENERATOR[7][4] = {{1, 1, 0, 1}, {1, 0, 1, 1}, {1, 0, 0, 0}, {0, 1, 1, 1},
                                      {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
static const int M_PARITY[3][7] = {{1, 0, 1, 0, 1, 0, 1}, {0, 1, 1, 0, 0, 1, 1},
                                   {0, 0, 0, 1, 1, 1, 1}};

/* ref:166-175 + :141-142 + :132-138.  Trailing <4 bits dropped. */
int64_t afsk_o_ecc_encode(const uint8_t *bits, int64_t n, uint8_t *out) {
    int64_t o = 0;
    for (int64_t i = 0; i < n - 3; i += 4) {
        for (int r = 0; r < 7; r++) {
            int acc = 0;
            for (int c = 0; c < 4; c++) acc += M_GENERATOR[r][c] * (bits[i + c] ? 1 : 0);
            out[o++] = (uint8_t)(acc % 2);
        }
    }
    return o;
}

/* ref:154-163 + :145-151 + :132-138.  Trailing <7 bits dropped. */
int64_t afsk_o_ecc_decode(const uint8_t *bits, int64_t n, uint8_t *out) {
    int64_t o = 0;
    for (int64_t i = 0; i < n - 6; i += 7) {
        int r[7], syn[3];
        for (int j = 0; j < 7; j++) r[j] = bits[i + j] ? 1 : 0;
        for (int a = 0; a < 3; a++) {
            int acc = 0;
            for (int j = 0; j < 7; j++) acc += M_PARITY[a][j] * r[j];
            syn[a] = acc % 2;
        }
        int error_pos = syn[2] * 4 + syn[1] * 2 + syn[0];     /* ref:147 */
        if (error_pos != 0) r[error_pos - 1] ^= 1;             /* ref:149-150 */
        out[o++] = (uint8_t)r[2];                              /* ref:151 */
        out[o++] = (uint8_t)r[4];
        out[o++] = (uint8_t)r[5];
        out[o++] = (uint8_t)r[6];
    }
    return o;
}

/* Number of codewords ECC.decode (ref:154-163) would correct: syndromes != 0 at ref:147-148. */
int64_t afsk_o_ecc_corrected(const uint8_t *bits, int64_t n) {
    int64_t count = 0;
    for (int64_t i = 0; i < n - 6; i += 7) {
        int syn[3];
        for (int a = 0; a < 3; a++) {
            int acc = 0;
            for (int j = 0; j < 7; j++) acc += M_PARITY[a][j] * (bits[i + j] ? 1 : 0);
            syn[a] = acc % 2;
        }
        if (syn[2] * 4 + syn[1] * 2 + syn[0] != 0) count++;
    }
    return count;
}

/* ref:393-399 MSB-first, trailing bits dropped. */
int64_t afsk_o_bits_to_bytes(const uint8_t *bits, int64_t n, uint8_t *out) {
    int64_t o = 0;
    for (int64_t i = 0; i <= n - 8; i += 8) {
        int v = 0;
        for (int j = 0; j < 8; j++) v = (v << 1) | (bits[i + j] ? 1 : 0);
        out[o++] = (uint8_t)v;
    }
    return o;
}

/* ref:446-450 '{0:08b}' per byte. */
int64_t afsk_o_bytes_to_bits(const uint8_t *data, int64_t n, uint8_t *out) {
    for (int64_t i = 0; i < n; i++)
        for (int j = 0; j < 8; j++) out[i * 8 + j] = (uint8_t)((data[i] >> (7 - j)) & 1);
    return n * 8;
}

/* -------------------------------------------------------------- Transmitter */

static void tones_from_bit_frames(int bit_frames, int16_t *space, int16_t *mark) {
    int h = bit_frames / 2, q = bit_frames / 4;
    for (int i = 0; i < h; i++) { space[i] = HI; space[h + i] = LO; }
    for (int i = 0; i < q; i++) {
        mark[i] = HI; mark[q + i] = LO; mark[2 * q + i] = HI; mark[3 * q + i] = LO;
    }
}

int64_t afsk_o_frame_count(int baud, int ts_cycles, int64_t nbytes) {
    if (baud <= 0 || SAMPLE_RATE % baud != 0 || SAMPLE_RATE % (2 * baud) != 0)
        return AFSK_O_ERR_INVALID_BAUD;
    int64_t space_len = 2 * (int64_t)(((double)SAMPLE_RATE / baud) / 2.0);
    int64_t mark_len = 4 * (int64_t)(((double)SAMPLE_RATE / (2 * baud)) / 2.0);
    int64_t ecc_bits = nbytes * 2 * 7; /* 2 nibbles per byte, 7 coded bits each */
    /* Every coded bit is a tone of its own length (mark and space lengths can
     * differ for bauds the receiver rejects); count the worst case exactly in
     * get_frames.  For valid bauds both equal bit_frames. */
    int64_t tone = mark_len > space_len ? mark_len : space_len;
    return (int64_t)ts_cycles * (mark_len + space_len) + mark_len + 3 * space_len +
           ecc_bits * tone + TAIL_SILENCE;
}

/* ref:452-469 */
int64_t afsk_o_get_frames(const uint8_t *data, int64_t nbytes, int baud, int ts_cycles,
                          int16_t *out, int64_t cap) {
    struct templates t;
    int rc = build_templates_baud(baud, &t);
    if (rc < 0) return rc;
    uint8_t *msg = (uint8_t *)malloc((size_t)(nbytes * 8 + 8));
    uint8_t *ecc = (uint8_t *)malloc((size_t)(nbytes * 14 + 14));
    int64_t nmsg = afsk_o_bytes_to_bits(data, nbytes, msg);        /* ref:454 */
    int64_t necc = afsk_o_ecc_encode(msg, nmsg, ecc);              /* ref:455 */
    int64_t o = 0;
    int64_t ret = 0;
#define EMIT(src, n)                                                   \
    do {                                                               \
        if (o + (n) > cap) { ret = AFSK_O_ERR_CAPACITY; goto done; }   \
        memcpy(out + o, (src), sizeof(int16_t) * (size_t)(n));         \
        o += (n);                                                      \
    } while (0)
    for (int i = 0; i < ts_cycles; i++) EMIT(t.training, t.training_len); /* ref:457-458 */
    EMIT(t.mark, t.mark_len);                                      /* ref:460 */
    for (int i = 0; i < 3; i++) EMIT(t.space, t.space_len);        /* ref:461-462 */
    for (int64_t i = 0; i < necc; i++) {                           /* ref:463-467 */
        if (ecc[i] == 0) EMIT(t.space, t.space_len);
        else EMIT(t.mark, t.mark_len);
    }
    if (o + TAIL_SILENCE > cap) { ret = AFSK_O_ERR_CAPACITY; goto done; }
    memset(out + o, 0, sizeof(int16_t) * TAIL_SILENCE);            /* ref:468 */
    o += TAIL_SILENCE;
    ret = o;
done:
#undef EMIT
    free(msg);
    free(ecc);
    free(t.space);
    return ret;
}

/* ref:239-244: for i in range(0, len-1, 2): emit frames[i] twice. */
int64_t afsk_o_wav_convert(const int16_t *in, int64_t n, int16_t *out) {
    int64_t o = 0;
    for (int64_t i = 0; i < n - 1; i += 2) {
        int16_t v = in[i];
        out[o++] = v;
        out[o++] = v;
    }
    return o;
}

/* ------------------------------------------------------------- live gate replay */

#define LISTEN_BLOCK 2048  /* ref:189, 209, 310 */

int32_t afsk_o_gate_stream(const int16_t *frames, int64_t len, int32_t amp_start_threshold,
                           int32_t amp_end_threshold, int32_t max_bursts, int32_t *burst_start,
                           int32_t *burst_len, int32_t *open_end) {
    const int64_t nb = len / LISTEN_BLOCK;
    int64_t b = 0;
    int32_t n = 0;
    *open_end = 0;
    while (n < max_bursts) {
        b += 1;                                                     /* ref:303 discard */
        while (b < nb &&
               !(afsk_o_get_amplitude(frames + b * LISTEN_BLOCK, LISTEN_BLOCK) > amp_start_threshold))
            b++;                                                    /* ref:304-310 */
        if (b >= nb) break;
        const int64_t start = b;                                    /* ref:307-309 */
        b++;
        int quiet = 0;
        while (b < nb) {                                            /* ref:313-318 */
            quiet = afsk_o_get_amplitude(frames + b * LISTEN_BLOCK, LISTEN_BLOCK) < amp_end_threshold;
            b++;
            if (quiet) break;
        }
        burst_start[n] = (int32_t)(start * LISTEN_BLOCK);
        burst_len[n] = (int32_t)((b - start) * LISTEN_BLOCK);
        n++;
        if (!quiet) { *open_end = 1; break; }
    }
    return n;
}

/* ------------------------------------------------------- deterministic noise */

static inline uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

void afsk_o_add_noise(int16_t *samples, int64_t len, uint32_t seed, uint32_t stream_idx,
                      int32_t scale_q24) {
    uint32_t key = hash32(seed ^ hash32(stream_idx + 0x9e3779b9U));
    for (int64_t t = 0; t < len; t++) {
        int32_t s = 0;
        for (uint32_t j = 0; j < 8; j++) {
            uint32_t h = hash32(key ^ ((uint32_t)t * 8u + j));
            s += (int32_t)(h & 0xffffu) + (int32_t)(h >> 16);
        }
        int64_t centred = (int64_t)s - 524280;                  /* 16 * 32767.5 */
        int64_t noise = (centred * (int64_t)scale_q24 + (1 << 23)) >> 24;
        int64_t v = (int64_t)samples[t] + noise;
        if (v > 32767) v = 32767;
        if (v < -32768) v = -32768;
        samples[t] = (int16_t)v;
    }
}

/* ------------------------------------------------------------ whole streams */

int afsk_o_demod_stream(const int16_t *frames, int64_t len, int bit_frames, int amp_end_threshold,
                        uint8_t *out_bytes, int32_t out_cap, int32_t *out_nbytes,
                        int32_t *out_nbits, int32_t *out_clock_idx, int32_t *out_term_frame,
                        int32_t *out_status) {
    return afsk_o_demod_stream_ex(frames, len, bit_frames, amp_end_threshold, out_bytes, out_cap,
                                  out_nbytes, out_nbits, out_clock_idx, out_term_frame,
                                  out_status, NULL, NULL, 0);
}

int afsk_o_demod_stream_ex(const int16_t *frames, int64_t len, int bit_frames,
                           int amp_end_threshold, uint8_t *out_bytes, int32_t out_cap,
                           int32_t *out_nbytes, int32_t *out_nbits, int32_t *out_clock_idx,
                           int32_t *out_term_frame, int32_t *out_status, int32_t *out_corrected,
                           int32_t *out_margins, int32_t margin_cap) {
    if (bit_frames <= 0 || bit_frames % 4 != 0 || bit_frames * 2 >= SYNC_WINDOW)
        return AFSK_O_ERR_INVALID_BAUD;
    struct templates t;
    int16_t *buf = (int16_t *)malloc(sizeof(int16_t) * (size_t)bit_frames * 4);
    t.space = buf;
    t.mark = buf + bit_frames;
    t.training = buf + 2 * bit_frames;
    tones_from_bit_frames(bit_frames, t.space, t.mark);
    memcpy(t.training, t.mark, sizeof(int16_t) * bit_frames);               /* ref:89-90 */
    memcpy(t.training + bit_frames, t.space, sizeof(int16_t) * bit_frames);
    t.space_len = t.mark_len = bit_frames;
    t.training_len = 2 * bit_frames;

    int64_t cap_bits = len / bit_frames + 8;
    uint8_t *bits = (uint8_t *)malloc((size_t)cap_bits * 2);
    uint8_t *dec = bits + cap_bits;
    int32_t ci = -1;
    int64_t term = -1;
    int64_t nbits = decode_bits_tpl(frames, len, bit_frames, &t, amp_end_threshold, bits,
                                    cap_bits, &ci, &term, out_margins, margin_cap);
    int rc = AFSK_O_OK;
    if (nbits < 0) {
        rc = (int)nbits;
    } else {
        int64_t ndec = afsk_o_ecc_decode(bits, nbits, dec);                 /* ref:425 */
        uint8_t *bytes = (uint8_t *)malloc((size_t)(ndec / 8 + 1));
        int64_t nbytes = afsk_o_bits_to_bytes(dec, ndec, bytes);            /* ref:426 */
        int64_t ncopy = nbytes < out_cap ? nbytes : out_cap;
        if (out_bytes && ncopy > 0) memcpy(out_bytes, bytes, (size_t)ncopy);
        free(bytes);
        if (out_corrected) *out_corrected = (int32_t)afsk_o_ecc_corrected(bits, nbits);
        *out_nbytes = (int32_t)nbytes;
        *out_nbits = (int32_t)nbits;
        *out_clock_idx = ci;
        *out_term_frame = (int32_t)term;
        *out_status = ci == -1 ? AFSK_O_ST_TOO_SHORT
                               : (nbits == 0 ? AFSK_O_ST_NO_DATA : AFSK_O_ST_OK); /* ref:422 */
    }
    free(bits);
    free(buf);
    return rc;
}

struct batch_job {
    const int16_t *samples;
    const int64_t *stream_offset;
    const int32_t *stream_len, *bit_frames;
    int32_t amp_end_threshold, begin, end;
    uint8_t *out_bytes;
    int32_t out_stride;
    int32_t *out_nbytes, *out_nbits, *out_clock_idx, *out_term_frame, *out_status;
    int rc;
};

static void *batch_worker(void *arg) {
    struct batch_job *j = (struct batch_job *)arg;
    for (int32_t s = j->begin; s < j->end; s++) {
        int rc = afsk_o_demod_stream(j->samples + j->stream_offset[s], j->stream_len[s],
                                     j->bit_frames[s], j->amp_end_threshold,
                                     j->out_bytes + (int64_t)s * j->out_stride, j->out_stride,
                                     &j->out_nbytes[s], &j->out_nbits[s], &j->out_clock_idx[s],
                                     &j->out_term_frame[s], &j->out_status[s]);
        if (rc < 0) { j->rc = rc; return NULL; }
    }
    return NULL;
}

int afsk_o_demod_batch(const int16_t *samples, const int64_t *stream_offset,
                       const int32_t *stream_len, const int32_t *bit_frames,
                       int32_t amp_end_threshold, int32_t n_streams, uint8_t *out_bytes,
                       int32_t out_stride, int32_t *out_nbytes, int32_t *out_nbits,
                       int32_t *out_clock_idx, int32_t *out_term_frame, int32_t *out_status,
                       int32_t n_threads) {
    if (n_threads < 1) n_threads = 1;
    if (n_threads > n_streams) n_threads = n_streams > 0 ? n_streams : 1;
    struct batch_job *jobs = (struct batch_job *)calloc((size_t)n_threads, sizeof(*jobs));
    pthread_t *tids = (pthread_t *)calloc((size_t)n_threads, sizeof(*tids));
    for (int t = 0; t < n_threads; t++) {
        struct batch_job *j = &jobs[t];
        j->samples = samples; j->stream_offset = stream_offset; j->stream_len = stream_len;
        j->bit_frames = bit_frames; j->amp_end_threshold = amp_end_threshold;
        j->begin = (int32_t)((int64_t)n_streams * t / n_threads);
        j->end = (int32_t)((int64_t)n_streams * (t + 1) / n_threads);
        j->out_bytes = out_bytes; j->out_stride = out_stride; j->out_nbytes = out_nbytes;
        j->out_nbits = out_nbits; j->out_clock_idx = out_clock_idx;
        j->out_term_frame = out_term_frame; j->out_status = out_status; j->rc = 0;
        if (n_threads == 1) batch_worker(j);
        else pthread_create(&tids[t], NULL, batch_worker, j);
    }
    int rc = 0;
    for (int t = 0; t < n_threads; t++) {
        if (n_threads > 1) pthread_join(tids[t], NULL);
        if (jobs[t].rc < 0) rc = jobs[t].rc;
    }
    free(jobs);
    free(tids);
    return rc;
}

int afsk_o_modulate_batch(const uint8_t *payload, int32_t payload_stride,
                          const int32_t *payload_len, const int32_t *bit_frames,
                          const int32_t *ts_cycles, const int64_t *stream_offset,
                          const int32_t *stream_len, int32_t n_streams, int32_t wav_quirk,
                          int16_t *samples) {
    for (int32_t s = 0; s < n_streams; s++) {
        int bf = bit_frames[s];
        if (bf <= 0 || SAMPLE_RATE % bf != 0) return AFSK_O_ERR_INVALID_BAUD;
        int baud = SAMPLE_RATE / bf;
        int64_t cap = afsk_o_frame_count(baud, ts_cycles[s], payload_len[s]);
        if (cap < 0) return (int)cap;
        int16_t *tmp = (int16_t *)malloc(sizeof(int16_t) * (size_t)(cap + 2));
        int64_t n = afsk_o_get_frames(payload + (int64_t)s * payload_stride, payload_len[s], baud,
                                      ts_cycles[s], tmp, cap);
        if (n < 0) { free(tmp); return (int)n; }
        if (wav_quirk) n = afsk_o_wav_convert(tmp, n, tmp);
        int16_t *dst = samples + stream_offset[s];
        int64_t L = stream_len[s];
        int64_t ncopy = n < L ? n : L;                 /* truncate or zero-pad to L */
        memcpy(dst, tmp, sizeof(int16_t) * (size_t)ncopy);
        if (L > ncopy) memset(dst + ncopy, 0, sizeof(int16_t) * (size_t)(L - ncopy));
        free(tmp);
    }
    return AFSK_O_OK;
}
