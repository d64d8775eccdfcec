// search_api.cpp -- SearchInit/SearchFree/SearchTask/SearchEnable/SearchCode of the reference
// (c/search_offline.cpp:74-117,205-292) on top of the gpsacq C ABI.  The file loop, the
// threshold and the report format are host work; every correlation runs on the GPU.
//
// Environment (the reference has no options besides its three globals):
//   GPSACQ_DEVICE=<n>       HIP device ordinal (default 0)
//   GPSACQ_REF_QUIRKS=1     reproduce the reference's fwd_buf overrun on PRN index 0
//   GPSACQ_BATCH_RUNS=<n>   runs (32 blocks each) searched per GPU batch (default 64)
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../include/gps_search.h"
#include "../../include/gpsacq.h"

static gpsacq_engine *g_engine = nullptr;
static bool g_busy[GPSACQ_NUM_SATS];

static int env_int(const char *name, int dflt) {
    const char *v = getenv(name);
    return (v && *v) ? atoi(v) : dflt;
}

int SearchInit() {
    if (g_engine) { gpsacq_destroy(g_engine); g_engine = nullptr; }
    gpsacq_params p;
    p.fc = FC;
    p.fs = FS;
    p.max_fo = max_fo;
    p.device = env_int("GPSACQ_DEVICE", 0);
    p.ref_quirks = env_int("GPSACQ_REF_QUIRKS", 0);
    int rc = gpsacq_create(&p, &g_engine);
    if (rc != GPSACQ_OK) fprintf(stderr, "gpsacq: %s\n", gpsacq_last_error());
    return rc;
}

void SearchFree() {
    gpsacq_destroy(g_engine);
    g_engine = nullptr;
}

void SearchEnable(int sv) {
    if (sv >= 0 && sv < GPSACQ_NUM_SATS) g_busy[sv] = false;
}

int SearchCode(int sv, int g1) { return gpsacq_search_code(sv, g1); }

void SearchTask(char *filename_1bit_bin) {
    FILE *fp = fopen(filename_1bit_bin, "rb");
    if (fp == NULL) {
        printf("can not open file!\n");
        return;
    }
    if (!g_engine) {
        fprintf(stderr, "gpsacq: SearchTask() before a successful SearchInit()\n");
        fclose(fp);
        return;
    }
    const size_t run_bytes = (size_t)GPSACQ_NUM_SATS * GPSACQ_BLOCK_BYTES;
    const int batch_runs = env_int("GPSACQ_BATCH_RUNS", 64) > 0 ? env_int("GPSACQ_BATCH_RUNS", 64) : 64;
    std::vector<unsigned char> buf(run_bytes * batch_runs);
    std::vector<gpsacq_peak> peaks((size_t)GPSACQ_NUM_SATS * batch_runs);
    int run_count = 0;
    for (;;) {
        size_t got = 0;
        while (got < buf.size()) {
            size_t r = fread(buf.data() + got, 1, buf.size() - got, fp);
            if (r == 0) break;
            got += r;
        }
        // a run is complete when all 32 of its Sample() calls got their 10 x 512 bytes (:135-140,239-244)
        const size_t runs = got / run_bytes;
        if (runs > 0) {
            const size_t nblk = runs * GPSACQ_NUM_SATS;
            int rc = gpsacq_search(g_engine, buf.data(), nblk, GPSACQ_BLOCK_BYTES, NULL, nblk, NULL, peaks.data());
            if (rc != GPSACQ_OK) {
                fprintf(stderr, "gpsacq: %s\n", gpsacq_last_error());
                fclose(fp);
                exit(2);
            }
            for (size_t r = 0; r < runs; r++, run_count++) {
                const gpsacq_peak *pk = &peaks[r * GPSACQ_NUM_SATS];
                int hit[GPSACQ_NUM_SATS], hit_count = 0;
                for (int sv = 0; sv < GPSACQ_NUM_SATS; sv++)
                    if (!(pk[sv].snr < 25)) hit[hit_count++] = sv;  // :248
                printf("%2d satellite: ", run_count);
                for (int i = 0; i < hit_count; i++) printf("%5d ", hit[i]);
                printf("\n");
                printf("%2d SNR(>=25): ", run_count);
                for (int i = 0; i < hit_count; i++) printf("%5.1f ", pk[hit[i]].snr);
                printf("\n");
                printf("%2d  lo_shift: ", run_count);
                for (int i = 0; i < hit_count; i++) printf("%5d ", pk[hit[i]].lo_shift);
                printf("\n");
                printf("%2d  ca_shift: ", run_count);
                for (int i = 0; i < hit_count; i++) printf("%5d ", pk[hit[i]].ca_shift);
                printf("\n");
                for (int sv = 0; sv < GPSACQ_NUM_SATS; sv++) printf("%2.0f ", pk[sv].snr);
                printf("\n\n");
            }
        }
        if (got < buf.size()) {
            printf("run out of file!\n");
            break;
        }
    }
    fclose(fp);
}
