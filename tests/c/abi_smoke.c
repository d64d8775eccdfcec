/* abi_smoke.c -- plain C99 client of include/gpsacq.h: proves the header compiles as C and the
 * library links with C linkage only.  Usage: abi_smoke <capture> <fc> <fs> [max_fo]
 * exit 0: searched and printed the best PRN of the first run; exit 2: no device (message on stderr). */
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
    return 0;
}
