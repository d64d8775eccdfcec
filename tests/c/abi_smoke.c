/* abi_smoke.c -- plain C99 client of include/gpsacq.h: proves the header compiles as C and the
 * library links with C linkage only.  Usage: abi_smoke <capture> <fc> <fs> [max_fo]
 * exit 0: searched and printed the best PRN of the first run, then ran the pipeline, the two multi-GPU entries and an 8-bit IQ
 * search and compared them with that search; exit 2: no device (message on stderr). */
#include <stdio.h>
#include <stdlib.h>

#include "gpsacq.h"

int main(int argc, char **argv) {
    if (argc < 4) { fprintf(stderr, "usage: %s capture fc fs [max_fo]\n", argv[0]); return 64; }
    gpsacq_params p;
    p.fc = atof(argv[2]); p.fs = atof(argv[3]); p.max_fo = argc > 4 ? atof(argv[4]) : 5000.0;
    p.device = 0; p.ref_quirks = 0;
    gpsacq_engine *e = NULL;
    int rc = gpsacq_create(&p, &e);
    if (rc != GPSACQ_OK) { fprintf(stderr, "gpsacq_create: %d: %s\n", rc, gpsacq_last_error()); return rc; }
    gpsacq_info info;
    gpsacq_get_info(e, &info);
    FILE *fp = fopen(argv[1], "rb");
    if (!fp) { fprintf(stderr, "cannot open %s\n", argv[1]); gpsacq_destroy(e); return 66; }
    static unsigned char bits[GPSACQ_NUM_SATS * GPSACQ_BLOCK_BYTES];
    size_t got = fread(bits, 1, sizeof bits, fp);
    fclose(fp);
    size_t nblk = got / GPSACQ_BLOCK_BYTES;
    gpsacq_peak peaks[GPSACQ_NUM_SATS];
    rc = gpsacq_search(e, bits, nblk, GPSACQ_BLOCK_BYTES, NULL, nblk, NULL, peaks);
    if (rc != GPSACQ_OK) { fprintf(stderr, "gpsacq_search: %d: %s\n", rc, gpsacq_last_error()); gpsacq_destroy(e); return rc; }
    int best = 0;
    for (size_t t = 1; t < nblk; t++) if (peaks[t].snr > peaks[best].snr) best = (int)t;
    gpsacq_handoff_t h;
    gpsacq_handoff(&peaks[best], p.fc, p.fs, 0.0, &h);
    printf("bins %d lags %d best sv %d snr %.1f lo_shift %d ca_shift %d doppler %.1f Hz\n", info.num_doppler, info.num_lags, best,
           peaks[best].snr, peaks[best].lo_shift, peaks[best].ca_shift, h.lo_dop_hz);
    /* the Doppler grid step and the multi-GPU entry (one device: the degenerate case) from plain C */
    rc = gpsacq_set_doppler_step(e, p.fs / GPSACQ_FFT_LEN / 2.0);
    if (rc != GPSACQ_OK) { fprintf(stderr, "gpsacq_set_doppler_step: %d: %s\n", rc, gpsacq_last_error()); gpsacq_destroy(e); return rc; }
    gpsacq_get_info(e, &info);
    gpsacq_destroy(e);
    {
        gpsacq_multi *m = NULL;
        gpsacq_task tasks[GPSACQ_NUM_SATS];
        gpsacq_peak mp[GPSACQ_NUM_SATS];
        int sv;
        rc = gpsacq_multi_create(&p, NULL, 1, &m);
        if (rc != GPSACQ_OK) { fprintf(stderr, "gpsacq_multi_create: %d: %s\n", rc, gpsacq_last_error()); return rc; }
        for (sv = 0; sv < GPSACQ_NUM_SATS; sv++) { tasks[sv].block = best; tasks[sv].prn = sv; }
        rc = gpsacq_multi_search_grid(m, bits, nblk, GPSACQ_BLOCK_BYTES, tasks, GPSACQ_NUM_SATS, mp);
        if (rc != GPSACQ_OK) { fprintf(stderr, "gpsacq_multi_search_grid: %d: %s\n", rc, gpsacq_last_error()); gpsacq_multi_destroy(m); return rc; }
        printf("half-bin grid: %d points of %.2f Hz; multi (1 device) block %d sv %d snr %.1f lo_shift %d ca_shift %d\n", info.num_doppler,
               info.doppler_step_hz, best, best % GPSACQ_NUM_SATS, mp[best % GPSACQ_NUM_SATS].snr, mp[best % GPSACQ_NUM_SATS].lo_shift,
               mp[best % GPSACQ_NUM_SATS].ca_shift);
        if (mp[best % GPSACQ_NUM_SATS].lo_shift != peaks[best].lo_shift || mp[best % GPSACQ_NUM_SATS].ca_shift != peaks[best].ca_shift) {
            fprintf(stderr, "multi-GPU entry disagrees with gpsacq_search\n");
            gpsacq_multi_destroy(m);
            return 70;
        }
        gpsacq_multi_destroy(m);
    }
    /* round 3: the pipeline (two batches in flight), the block decomposition over two engines sharing device 0, and an
     * 8-bit IQ search -- all from plain C */
    {
        gpsacq_engine *e2 = NULL;
        const gpsacq_peak *pk0 = NULL, *pk1 = NULL;
        size_t n0 = 0, n1 = 0, half = nblk / 2, t;
        unsigned char *b0, *b1;
        rc = gpsacq_create(&p, &e2);
        if (rc != GPSACQ_OK) { fprintf(stderr, "gpsacq_create: %d: %s\n", rc, gpsacq_last_error()); return rc; }
        if (gpsacq_reserve(e2, nblk) != GPSACQ_OK) { fprintf(stderr, "gpsacq_reserve: %s\n", gpsacq_last_error()); return 71; }
        b0 = gpsacq_pipe_buffer(e2, 0, half * GPSACQ_BLOCK_BYTES);
        b1 = gpsacq_pipe_buffer(e2, 1, (nblk - half) * GPSACQ_BLOCK_BYTES);
        if (!b0 || !b1) { fprintf(stderr, "gpsacq_pipe_buffer: %s\n", gpsacq_last_error()); return 71; }
        for (t = 0; t < half * GPSACQ_BLOCK_BYTES; t++) b0[t] = bits[t];
        for (t = 0; t < (nblk - half) * GPSACQ_BLOCK_BYTES; t++) b1[t] = bits[half * GPSACQ_BLOCK_BYTES + t];
        if (gpsacq_pipe_submit(e2, 0, half, GPSACQ_BLOCK_BYTES, NULL) != GPSACQ_OK ||
            gpsacq_pipe_submit(e2, 1, nblk - half, GPSACQ_BLOCK_BYTES, NULL) != GPSACQ_OK ||
            gpsacq_pipe_collect(e2, 0, &pk0, &n0) != GPSACQ_OK || gpsacq_pipe_collect(e2, 1, &pk1, &n1) != GPSACQ_OK) {
            fprintf(stderr, "gpsacq_pipe: %s\n", gpsacq_last_error());
            return 71;
        }
        /* batch 0 is blocks 0.. against PRN index = block: the same tasks as the first half of gpsacq_search above */
        for (t = 0; t < half; t++)
            if (pk0[t].ca_shift != peaks[t].ca_shift || pk0[t].lo_shift != peaks[t].lo_shift || pk0[t].snr != peaks[t].snr) {
                fprintf(stderr, "pipeline disagrees with gpsacq_search at block %d\n", (int)t);
                return 72;
            }
        printf("pipeline: %d + %d peaks\n", (int)n0, (int)n1);
        gpsacq_destroy(e2);
    }
    if (nblk == GPSACQ_NUM_SATS) {
        gpsacq_multi *m = NULL;
        int32_t dev[2] = {0, 0};
        gpsacq_peak all[GPSACQ_NUM_SATS], bestp[GPSACQ_NUM_SATS];
        size_t t;
        rc = gpsacq_multi_create(&p, dev, 2, &m);
        if (rc != GPSACQ_OK) { fprintf(stderr, "gpsacq_multi_create(0,0): %d: %s\n", rc, gpsacq_last_error()); return rc; }
        rc = gpsacq_multi_search_blocks(m, bits, 1, GPSACQ_BLOCK_BYTES, all, bestp);
        if (rc != GPSACQ_OK) { fprintf(stderr, "gpsacq_multi_search_blocks: %d: %s\n", rc, gpsacq_last_error()); gpsacq_multi_destroy(m); return rc; }
        for (t = 0; t < GPSACQ_NUM_SATS; t++)
            if (all[t].ca_shift != peaks[t].ca_shift || bestp[t].ca_shift != peaks[t].ca_shift || bestp[t].max_pwr != peaks[t].max_pwr) {
                fprintf(stderr, "block decomposition disagrees with gpsacq_search at PRN index %d\n", (int)t);
                gpsacq_multi_destroy(m);
                return 73;
            }
        printf("multi blocks (2 engines on device 0): best sv %d snr %.1f max_pwr %.4g\n", best, bestp[best].snr, bestp[best].max_pwr);
        {
            double enq = -1.0, tot = -1.0;
            int64_t ar = -1;
            if (gpsacq_multi_last_call_ms(m, &enq, &tot, &ar) != GPSACQ_OK || !(enq > 0.0) || !(tot >= enq) || ar != 0) {
                fprintf(stderr, "gpsacq_multi_last_call_ms: enqueue %.3f total %.3f all-reduces %d\n", enq, tot, (int)ar);
                gpsacq_multi_destroy(m);
                return 76;
            }
        }
        gpsacq_multi_destroy(m);
    }
    {
        /* an rtl-sdr style buffer made from the first block: I = 128 +- 40 by the sample's sign, Q = 128: its real part has the
         * capture's signs, so the IQ search of it must equal the 1-bit search of block 0 */
        gpsacq_engine *e3 = NULL;
        static unsigned char iq[2 * 8 * GPSACQ_BLOCK_BYTES];
        gpsacq_iq8_input in;
        gpsacq_task t0;
        gpsacq_peak a, b;
        size_t n;
        rc = gpsacq_create(&p, &e3);
        if (rc != GPSACQ_OK) { fprintf(stderr, "gpsacq_create: %d: %s\n", rc, gpsacq_last_error()); return rc; }
        for (n = 0; n < (size_t)8 * GPSACQ_BLOCK_BYTES; n++) {
            iq[2 * n] = (unsigned char)(((bits[n >> 3] >> (n & 7)) & 1) ? 88 : 168);
            iq[2 * n + 1] = 128;
        }
        in.format = GPSACQ_IQ_U8; in.remove_dc = 0; in.mean_i = in.mean_q = 0.0; in.mix_hz = 0.0; in.fs = 0.0;
        in.first_sample = 0; in.total_samples = 0; in.multibit = 0; in.reserved = 0;
        t0.block = 0; t0.prn = best % GPSACQ_NUM_SATS;
        if (gpsacq_search_iq8(e3, &in, iq, 1, sizeof iq, &t0, 1, NULL, &a) != GPSACQ_OK ||
            gpsacq_search(e3, bits, 1, GPSACQ_BLOCK_BYTES, &t0, 1, NULL, &b) != GPSACQ_OK) {
            fprintf(stderr, "gpsacq_search_iq8: %s\n", gpsacq_last_error());
            return 74;
        }
        if (a.snr != b.snr || a.ca_shift != b.ca_shift || a.lo_shift != b.lo_shift) { fprintf(stderr, "IQ search disagrees with the 1-bit search\n"); return 75; }
        printf("iq8: block 0 sv %d snr %.1f == 1-bit\n", t0.prn, a.snr);
        gpsacq_destroy(e3);
    }
    return 0;
}
