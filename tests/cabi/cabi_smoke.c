/* cabi_smoke.c -- a plain C caller of include/afsk_amd.h (no Python, no torch, no HIP headers).
 *
 * Builds one clean 1200-baud stream the way Transmitter.__getFrames does (afskmodem.py:452-469:
 * training cycles, terminator mark/space/space/space, Hamming(7,4)-coded payload bits MSB first,
 * 4800 zero frames), hands it to afsk_demod_batch_host and checks the decoded bytes.
 * Exit code: 0 = decoded and equal, 3 = library reports "no device" (expected on a CPU-only
 * box), anything else = failure.  Built and run by tests/test_host_api.py and
 * tests/test_gpu_parity.py.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "afsk_amd.h"

static int16_t *g_out;
static size_t g_n;

static void tone(int mark, int bf) {
    /* space = hi half, lo half (:68-77); mark = hi, lo, hi, lo quarters (:80-85) */
    for (int j = 0; j < bf; j++) {
        int hi = mark ? (((j / (bf / 4)) & 1) == 0) : (j < bf / 2);
        g_out[g_n++] = hi ? 32767 : -32768;
    }
}

int main(void) {
    const int baud = 1200, bf = AFSK_SAMPLE_RATE / baud;
    const char *msg = "C-ABI ok";
    const int nbytes = (int)strlen(msg);
    const int ts_cycles = (int)(baud * 0.5 / 2);                       /* :438 */
    g_out = (int16_t *)calloc((size_t)(2 * ts_cycles + 4 + 14 * nbytes) * bf + AFSK_TAIL_SILENCE, 2);
    for (int c = 0; c < ts_cycles; c++) { tone(1, bf); tone(0, bf); }  /* :457-458 */
    tone(1, bf); tone(0, bf); tone(0, bf); tone(0, bf);                /* :460-462 */
    for (int b = 0; b < nbytes; b++) {
        for (int half = 0; half < 2; half++) {
            int nib = half == 0 ? ((unsigned char)msg[b] >> 4) : (msg[b] & 15);
            int d1 = (nib >> 3) & 1, d2 = (nib >> 2) & 1, d3 = (nib >> 1) & 1, d4 = nib & 1;
            int cw[7] = {d1 ^ d2 ^ d4, d1 ^ d3 ^ d4, d1, d2 ^ d3 ^ d4, d2, d3, d4};   /* :115-123 */
            for (int k = 0; k < 7; k++) tone(cw[k], bf);
        }
    }
    g_n += AFSK_TAIL_SILENCE;                                          /* :468, already zero */

    int64_t off = 0;
    int32_t len = (int32_t)g_n, bfv = bf, nb = 0, nbits = 0, ci = 0, tf = 0, st = 0;
    uint8_t out[64];
    memset(out, 0, sizeof out);
    printf("afsk_version %d, devices %d, stream of %d samples\n", afsk_version(), afsk_device_count(), len);
    int rc = afsk_demod_batch_host(g_out, (int64_t)g_n, &off, &len, &bfv, 14000, 1, out, (int32_t)sizeof out,
                                   &nb, &nbits, &ci, &tf, &st);
    if (rc != AFSK_OK) {
        char err[256];
        afsk_last_error(err, (int)sizeof err);
        printf("afsk_demod_batch_host -> %d: %s\n", rc, err);
        return rc == AFSK_E_NO_DEVICE ? 3 : 1;
    }
    printf("status %d clock_idx %d term_frame %d nbits %d nbytes %d payload '%.*s'\n", st, ci, tf, nbits, nb,
           nb, (const char *)out);
    if (st != AFSK_ST_OK || nb != nbytes || memcmp(out, msg, (size_t)nbytes) != 0) return 1;
    /* the gather entry on the same stream */
    const int16_t *ptrs[1] = {g_out};
    memset(out, 0, sizeof out);
    rc = afsk_demod_streams_host(ptrs, &len, &bfv, 14000, 1, out, (int32_t)sizeof out, &nb, &nbits, &ci, &tf, &st);
    if (rc != AFSK_OK || nb != nbytes || memcmp(out, msg, (size_t)nbytes) != 0) return 1;
    if (afsk_host_scratch_release() != AFSK_OK) return 1;
    printf("OK\n");
    return 0;
}
