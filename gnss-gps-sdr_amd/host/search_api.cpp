// search_api.cpp -- SearchInit/SearchFree/SearchTask/SearchEnable/SearchCode of the reference
// (c/search_offline.cpp:74-117,205-292) on top of the gpsacq C ABI.  The file loop, the
// threshold and the report format are host work; every correlation runs on the GPU(s).
//
// Environment (the reference has no options besides its three globals):
//   GPSACQ_DEVICE=<n>       HIP device ordinal (default 0)
//   GPSACQ_DEVICES=a,b,..   several devices: each batch of runs is split into contiguous ranges,
//                           one host thread and one engine per device (runs are independent; the
//                           report is printed in file order).  Overrides GPSACQ_DEVICE.
//   GPSACQ_REF_QUIRKS=1     reproduce the reference's fwd_buf overrun on PRN index 0
//   GPSACQ_BATCH_RUNS=<n>   runs (32 blocks each) searched per device per batch (default 64)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/gps_search.h"
#include "../../include/gpsacq.h"

static std::vector<gpsacq_engine *> g_engines;
static bool g_busy[GPSACQ_NUM_SATS];
static int g_status = 0;  // status of the last SearchTask(): 0, or the gpsacq error that stopped it (SearchStatus())

static int env_int(const char *name, int dflt) {
    const char *v = getenv(name);
    return (v && *v) ? atoi(v) : dflt;
}

static std::vector<int> device_list() {
    std::vector<int> devs;
    const char *v = getenv("GPSACQ_DEVICES");
    if (v && *v) {
        std::string s(v);
        size_t pos = 0;
        while (pos <= s.size()) {
            size_t comma = s.find(',', pos);
            if (comma == std::string::npos) comma = s.size();
            if (comma > pos) devs.push_back(atoi(s.substr(pos, comma - pos).c_str()));
            pos = comma + 1;
        }
    }
    if (devs.empty()) devs.push_back(env_int("GPSACQ_DEVICE", 0));
    return devs;
}

void SearchFree() {
    for (gpsacq_engine *e : g_engines) gpsacq_destroy(e);
    g_engines.clear();
}

int SearchInit() {
    SearchFree();
    for (int dev : device_list()) {
        gpsacq_params p;
        p.fc = FC;
        p.fs = FS;
        p.max_fo = max_fo;
        p.device = dev;
        p.ref_quirks = env_int("GPSACQ_REF_QUIRKS", 0);
        gpsacq_engine *e = nullptr;
        int rc = gpsacq_create(&p, &e);
        if (rc != GPSACQ_OK) {
            fprintf(stderr, "gpsacq: %s\n", gpsacq_last_error());
            SearchFree();
            return rc;
        }
        g_engines.push_back(e);
    }
    return 0;
}

void SearchEnable(int sv) {
    if (sv >= 0 && sv < GPSACQ_NUM_SATS) g_busy[sv] = false;
}

int SearchCode(int sv, int g1) { return gpsacq_search_code(sv, g1); }

int SearchStatus() { return g_status; }

void SearchTask(char *filename_1bit_bin) {
    g_status = 0;
    FILE *fp = fopen(filename_1bit_bin, "rb");
    if (fp == NULL) {
        printf("can not open file!\n");
        return;
    }
    if (g_engines.empty()) {
        fprintf(stderr, "gpsacq: SearchTask() before a successful SearchInit()\n");
        fclose(fp);
        g_status = GPSACQ_ERR_ARG;
        return;
    }
    const size_t run_bytes = (size_t)GPSACQ_NUM_SATS * GPSACQ_BLOCK_BYTES;
    const size_t n_dev = g_engines.size();
    const int batch_env = env_int("GPSACQ_BATCH_RUNS", 64);
    const size_t batch_runs = (size_t)(batch_env > 0 ? batch_env : 64) * n_dev;
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
            // contiguous run ranges, one per device; a run always starts at PRN index 0, so the
            // reference schedule (block t <-> PRN t % 32) holds inside every range
            std::vector<int> rcs(n_dev, GPSACQ_OK);
            std::vector<std::string> errs(n_dev);
            std::vector<std::thread> workers;
            for (size_t d = 0; d < n_dev; d++) {
                const size_t first = runs * d / n_dev, last = runs * (d + 1) / n_dev;
                if (last == first) continue;
                auto job = [&, d, first, last]() {
                    const size_t nblk = (last - first) * GPSACQ_NUM_SATS;
                    rcs[d] = gpsacq_search(g_engines[d], buf.data() + first * run_bytes, nblk, GPSACQ_BLOCK_BYTES, NULL, nblk,
                                           NULL, peaks.data() + first * GPSACQ_NUM_SATS);
                    if (rcs[d] != GPSACQ_OK) errs[d] = gpsacq_last_error();
                };
                if (n_dev == 1) job();
                else workers.emplace_back(job);
            }
            for (std::thread &w : workers) w.join();
            for (size_t d = 0; d < n_dev; d++)
                if (rcs[d] != GPSACQ_OK) {  // a library function does not exit(): report, stop, leave the status for the caller
                    fprintf(stderr, "gpsacq: %s\n", errs[d].c_str());
                    g_status = rcs[d];
                }
            if (g_status != 0) break;
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
            fflush(stdout);  // the reference prints run by run; here a batch of runs at a time
        }
        if (got < buf.size()) {
            printf("run out of file!\n");
            break;
        }
    }
    fclose(fp);
}
