// search_api.cpp -- SearchInit/SearchFree/SearchTask/SearchEnable/SearchCode of the reference
// (c/search_offline.cpp:74-117,205-292) on top of the gpsacq C ABI.  The file loop, the
// threshold and the report format are host work; every correlation runs on the GPU(s).
//
// SearchTask() is a pipeline: batches of whole runs are read straight into pinned staging buffers
// (gpsacq_pipe_buffer), handed to the engine(s) without waiting (gpsacq_pipe_submit: upload on a second
// stream, search behind it) and collected one batch later, so the fread of batch k+1 and the printf of
// batch k-1 overlap the search of batch k.  The first batches are short (1, 2, 4, ... runs) so that the
// first run's report leaves as early as the reference's does; every run is printed and flushed as soon
// as its batch lands (the reference prints run by run, :264-287).
//
// Environment (the reference has no options besides its three globals):
//   GPSACQ_DEVICE=<n>       HIP device ordinal (default 0)
//   GPSACQ_DEVICES=a,b,..   several devices: each batch of runs is split into contiguous ranges, one engine
//                           per device, all fed from this one thread (submits do not block; runs are
//                           independent; the report is printed in file order).  Overrides GPSACQ_DEVICE.
//   GPSACQ_REF_QUIRKS=1     reproduce the reference's fwd_buf overrun on PRN index 0
//   GPSACQ_BATCH_RUNS=<n>   most runs (32 blocks each) searched per device per batch (default 64; 16 for IQ input)
//   GPSACQ_INPUT=bits|iq_u8|iq_s8   the capture file's format: gps_test's own 1-bit stream (default), or the 8-bit IQ
//                           file of an rtl-sdr (uint8, offset 128) / HackRF (int8) -- README.md:83-115's flow without the
//                           MATLAB step (proc_rtl_bin_for_gps.m, proc_hackrf_bin_for_gps.m): mean removal, mixer and
//                           sign happen on the GPU inside the forward transform
//   GPSACQ_MIX_HZ=<f>       IQ input: mix the baseband capture up to this real IF first (proc_rtl_bin_for_gps.m:31-47,
//                           fc = 0.62e6 there); 0 / unset: take the real part (:12-26).  FC should name the same IF.
//   GPSACQ_IQ_KEEP_DC=1     IQ input: skip `y = y - mean(y)` (the scripts always remove it)
//   GPSACQ_IQ_MULTIBIT=1    IQ input: keep the samples' amplitude instead of their sign (no reference counterpart: gps_test reads
//                           1-bit files only); spares the 1-bit quantisation loss
//   GPSACQ_IQ_COMPLEX=1     IQ input: the capture is at baseband already, I + jQ being what Sample() puts in fwd_buf -- the int8
//                           file c/conv_1bit_bin_to_hackrf_bin.cpp writes (GPSACQ_INPUT=iq_s8 GPSACQ_IQ_KEEP_DC=1): transformed as it
//                           is, no LO; FC is not used and GPSACQ_MIX_HZ, if set, turns the samples by that frequency first
//   GPSACQ_SUM_THREADS=<n>  IQ input: host threads that sum the capture for its mean (default 8, at most the cores present);
//   GPSACQ_SUMS_ON_GPU=1    ... or read the file through a buffer and sum it on the GPU (the fallback when it cannot be mapped)
//   GPSACQ_NO_MMAP=1        read the capture with fread only (what happens anyway when it is not a regular file); by default it is
//                           mapped and the batches are copied into the staging buffers by GPSACQ_SUM_THREADS threads.  The file's
//                           size is taken again (fstat) before every batch, so a capture that shrinks between batches ends the
//                           search like the reference's short fread does; one truncated DURING a batch's copy raises SIGBUS --
//                           the caveat of every mapping -- and GPSACQ_NO_MMAP=1 is the mode for a file another process rewrites
//   GPSACQ_TRACE=1          wall-clock split of SearchInit / SearchTask on stderr
#include <sys/mman.h>
#include <sys/stat.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <string>
#include <thread>
#include <vector>

#include "../../include/gps_search.h"
#include "../../include/gpsacq.h"

static std::vector<gpsacq_engine *> g_engines;
static bool g_busy[GPSACQ_NUM_SATS];
static int g_status = 0;  // status of the last SearchTask(): 0, or the gpsacq error that stopped it (SearchStatus())
static double g_init_ms = 0;

typedef std::chrono::steady_clock Clock;
static double ms_since(Clock::time_point t0) { return std::chrono::duration<double, std::milli>(Clock::now() - t0).count(); }

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
    const Clock::time_point t0 = Clock::now();
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
    g_init_ms = ms_since(t0);
    return 0;
}

void SearchEnable(int sv) {
    if (sv >= 0 && sv < GPSACQ_NUM_SATS) g_busy[sv] = false;
}

int SearchCode(int sv, int g1) { return gpsacq_search_code(sv, g1); }

int SearchStatus() { return g_status; }

// the five lines + blank of one run (:264-287)
static void print_run(int run, const gpsacq_peak *pk) {
    int hit[GPSACQ_NUM_SATS], hit_count = 0;
    for (int sv = 0; sv < GPSACQ_NUM_SATS; sv++)
        if (!(pk[sv].snr < 25)) hit[hit_count++] = sv;  // :248
    printf("%2d satellite: ", run);
    for (int i = 0; i < hit_count; i++) printf("%5d ", hit[i]);
    printf("\n");
    printf("%2d SNR(>=25): ", run);
    for (int i = 0; i < hit_count; i++) printf("%5.1f ", pk[hit[i]].snr);
    printf("\n");
    printf("%2d  lo_shift: ", run);
    for (int i = 0; i < hit_count; i++) printf("%5d ", pk[hit[i]].lo_shift);
    printf("\n");
    printf("%2d  ca_shift: ", run);
    for (int i = 0; i < hit_count; i++) printf("%5d ", pk[hit[i]].ca_shift);
    printf("\n");
    for (int sv = 0; sv < GPSACQ_NUM_SATS; sv++) printf("%2.0f ", pk[sv].snr);
    printf("\n\n");
}

static void fail_with(const char *what) {
    fprintf(stderr, "gpsacq: %s%s\n", what, gpsacq_last_error());
    g_status = GPSACQ_ERR_DEVICE;
}

static size_t read_fully(FILE *fp, unsigned char *dst, size_t want) {
    size_t got = 0;
    while (got < want) {
        size_t r = fread(dst + got, 1, want - got, fp);
        if (r == 0) break;
        got += r;
    }
    return got;
}

// `y - mean(y)` needs the two integer sums of the whole IQ capture before the first block can be converted: computed on the host
// cores straight from the page cache (mmap, a few threads, each a contiguous range), which is several times faster than reading
// the file through a buffer and summing it on the GPU -- and exact either way (gpsacq_iq8_accumulate_sums is the fallback when
// the file cannot be mapped).
__attribute__((optimize("O3", "tree-vectorize"))) static void iq_sums_range(const unsigned char *p, size_t n_pairs, int is_signed, int64_t *out) {
    int64_t si = 0, sq = 0;
    size_t k = 0;
    while (k < n_pairs) {
        const size_t m = n_pairs - k < 16384 ? n_pairs - k : 16384;  // 32-bit partial sums cannot overflow over 16384 samples
        int32_t a = 0, b = 0;
        const unsigned char *q = p + 2 * k;
        if (is_signed) for (size_t j = 0; j < m; j++) { a += (int8_t)q[2 * j]; b += (int8_t)q[2 * j + 1]; }
        else for (size_t j = 0; j < m; j++) { a += q[2 * j]; b += q[2 * j + 1]; }
        si += a;
        sq += b;
        k += m;
    }
    if (!is_signed) {  // offset 128 (rtl-sdr)
        si -= 128 * (int64_t)n_pairs;
        sq -= 128 * (int64_t)n_pairs;
    }
    out[0] = si;
    out[1] = sq;
}
static int host_threads(uint64_t items_of_work) {
    unsigned hw = std::thread::hardware_concurrency();
    const int want = env_int("GPSACQ_SUM_THREADS", 8);
    int nt = want < 1 ? 1 : want;
    if (hw > 0 && (unsigned)nt > hw) nt = (int)hw;
    if ((uint64_t)nt > items_of_work + 1) nt = (int)(items_of_work + 1);
    return nt;
}
// a batch of an 8-bit IQ file is tens of megabytes: copied from the mapped file into the pinned staging buffer by a few threads
// (one thread's fread moves ~9 GB/s, which left the GPU waiting for the file)
static void copy_parallel(unsigned char *dst, const unsigned char *src, size_t n) {
    const int nt = host_threads(n >> 22);  // a thread per 4 MB at least
    if (nt <= 1) {
        memcpy(dst, src, n);
        return;
    }
    std::vector<std::thread> workers;
    for (int t = 1; t < nt; t++)
        workers.emplace_back([=] {
            const size_t a = n * (size_t)t / (size_t)nt, b = n * (size_t)(t + 1) / (size_t)nt;
            memcpy(dst + a, src + a, b - a);
        });
    memcpy(dst, src, n / (size_t)nt);
    for (std::thread &w : workers) w.join();
}
static bool iq_sums_mapped(const unsigned char *base, uint64_t n_pairs, int is_signed, int64_t sums[2]) {
    const int nt = host_threads(n_pairs / 65536);
    std::vector<int64_t> part((size_t)nt * 2, 0);
    std::vector<std::thread> workers;
    for (int t = 1; t < nt; t++)
        workers.emplace_back([&, t] {
            const uint64_t a = n_pairs * (uint64_t)t / (uint64_t)nt, b = n_pairs * (uint64_t)(t + 1) / (uint64_t)nt;
            iq_sums_range(base + 2 * a, (size_t)(b - a), is_signed, &part[(size_t)t * 2]);
        });
    iq_sums_range(base, (size_t)(n_pairs / (uint64_t)nt), is_signed, &part[0]);
    for (std::thread &w : workers) w.join();
    sums[0] = sums[1] = 0;
    for (int t = 0; t < nt; t++) {
        sums[0] += part[(size_t)t * 2];
        sums[1] += part[(size_t)t * 2 + 1];
    }
    return true;
}

void SearchTask(char *filename_1bit_bin) {
    g_status = 0;
    const Clock::time_point t_start = Clock::now();
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
    const bool trace = env_int("GPSACQ_TRACE", 0) != 0;
    double ms_sums = 0, ms_read = 0, ms_submit = 0, ms_wait = 0, ms_print = 0;
    // the capture mapped read-only when it can be (a regular file): the IQ mean and the batch copies then run on several threads;
    // otherwise (a pipe, GPSACQ_NO_MMAP=1) everything goes through fread as before
    struct Mapping {  // unmapped on every way out of this function
        const unsigned char *p = NULL;
        size_t len = 0;
        ~Mapping() {
            if (p) munmap((void *)p, len);
        }
    } mapping;
    size_t map_pos = 0;
    if (env_int("GPSACQ_NO_MMAP", 0) == 0 && fseek(fp, 0, SEEK_END) == 0) {
        const long long fsize = ftell(fp);
        fseek(fp, 0, SEEK_SET);
        if (fsize > 0) {
            void *m = mmap(NULL, (size_t)fsize, PROT_READ, MAP_SHARED, fileno(fp), 0);
            if (m != MAP_FAILED) {
                mapping.p = (const unsigned char *)m;
                mapping.len = (size_t)fsize;
                (void)madvise(m, mapping.len, MADV_SEQUENTIAL);
            }
        }
    }
    const unsigned char *const map = mapping.p;
    const size_t map_len = mapping.len;

    // ---- input format -------------------------------------------------------------------------------------------
    const char *fmt = getenv("GPSACQ_INPUT");
    bool iq = false;
    gpsacq_iq8_input iqin;
    memset(&iqin, 0, sizeof iqin);
    if (fmt && *fmt && strcmp(fmt, "bits") != 0) {
        if (strcmp(fmt, "iq_u8") == 0) iqin.format = GPSACQ_IQ_U8;
        else if (strcmp(fmt, "iq_s8") == 0) iqin.format = GPSACQ_IQ_S8;
        else {
            fprintf(stderr, "gpsacq: GPSACQ_INPUT=%s is not one of bits, iq_u8, iq_s8\n", fmt);
            fclose(fp);
            g_status = GPSACQ_ERR_ARG;
            return;
        }
        iq = true;
        const char *mix = getenv("GPSACQ_MIX_HZ");
        iqin.mix_hz = (mix && *mix) ? atof(mix) : 0.0;
        iqin.fs = FS;
        iqin.remove_dc = env_int("GPSACQ_IQ_KEEP_DC", 0) ? 0 : 1;
        iqin.multibit = env_int("GPSACQ_IQ_COMPLEX", 0) ? GPSACQ_SAMPLES_COMPLEX : env_int("GPSACQ_IQ_MULTIBIT", 0) ? GPSACQ_SAMPLES_REAL : GPSACQ_SAMPLES_SIGN;
    }
    const size_t block_bytes = iq ? (size_t)GPSACQ_BLOCK_BYTES * 16 : (size_t)GPSACQ_BLOCK_BYTES;  // one Sample(): 40960 samples
    const size_t run_bytes = (size_t)GPSACQ_NUM_SATS * block_bytes;
    const size_t n_dev = g_engines.size();
    // per device per batch; an IQ run is 16 times the bytes of a 1-bit run (2.6 MB), so fewer of them fill a staging buffer
    const int batch_dflt = iq ? 16 : 64, batch_env = env_int("GPSACQ_BATCH_RUNS", batch_dflt);
    size_t max_runs = (size_t)(batch_env > 0 ? batch_env : batch_dflt);
    // a mapped file's length is known: a short capture (the bundled gps_sig_tmp.bin has 12 runs) gets staging buffers of its own
    // size, not of the largest batch (pinned allocations are the most expensive item of a short SearchTask)
    if (map && map_len / run_bytes + 1 < max_runs) max_runs = map_len / run_bytes + 1;

    if (iq) {
        // `y = y - mean(y)` is the mean of the WHOLE capture (proc_rtl_bin_for_gps.m:17): one pass for the two integer sums
        const Clock::time_point t0 = Clock::now();
        fseek(fp, 0, SEEK_END);
        const long long fsize = ftell(fp);
        fseek(fp, 0, SEEK_SET);
        iqin.total_samples = (uint64_t)(fsize / 2);
        if (iqin.remove_dc && iqin.total_samples > 0) {
            int64_t sums[2] = {0, 0};
            const bool mapped = map && env_int("GPSACQ_SUMS_ON_GPU", 0) == 0 && iq_sums_mapped(map, iqin.total_samples, iqin.format == GPSACQ_IQ_S8, sums);
            std::vector<unsigned char> chunk(mapped ? 0 : (size_t)64 << 20);
            uint64_t left = mapped ? 0 : iqin.total_samples;
            while (left > 0) {
                const size_t want = (size_t)(left * 2 < chunk.size() ? left * 2 : chunk.size());
                const size_t got = read_fully(fp, chunk.data(), want);
                if (got < 2) break;
                if (gpsacq_iq8_accumulate_sums(g_engines[0], chunk.data(), got / 2, iqin.format, sums) != GPSACQ_OK) {
                    fail_with("");
                    fclose(fp);
                    return;
                }
                left -= got / 2;
                if (got < want) break;
            }
            iqin.mean_i = (double)sums[0] / (double)iqin.total_samples;  // exact sums; MATLAB's mean() of integer-valued doubles
            iqin.mean_q = (double)sums[1] / (double)iqin.total_samples;
            fseek(fp, 0, SEEK_SET);
        }
        ms_sums = ms_since(t0);
    }

    // ---- the pipeline -------------------------------------------------------------------------------------------
    const Clock::time_point t_setup = Clock::now();
    for (gpsacq_engine *e : g_engines) {  // scratch and staging for the largest batch once, not regrown as the batches ramp up
        bool ok = gpsacq_reserve(e, max_runs * GPSACQ_NUM_SATS) == GPSACQ_OK;
        for (int sl = 0; ok && sl < GPSACQ_PIPE_SLOTS; sl++) ok = gpsacq_pipe_buffer(e, sl, max_runs * run_bytes) != NULL;
        if (!ok) {
            fail_with("");
            fclose(fp);
            return;
        }
    }
    const double ms_setup = ms_since(t_setup);
    struct Batch {
        int slot;
        std::vector<size_t> runs;  // per device
    };
    std::deque<Batch> inflight;
    int run_count = 0, slot = 0;
    size_t ramp = 1;  // runs per device of the next batch: 1, 2, 4, ... max_runs
    uint64_t sample_pos = 0;
    bool eof = false;

    auto drain_one = [&]() -> bool {  // collect + print the oldest batch
        Batch b = inflight.front();
        inflight.pop_front();
        for (size_t d = 0; d < n_dev; d++) {
            if (b.runs[d] == 0) continue;
            const gpsacq_peak *pk = nullptr;
            size_t n = 0;
            Clock::time_point t0 = Clock::now();
            if (gpsacq_pipe_collect(g_engines[d], b.slot, &pk, &n) != GPSACQ_OK) {
                fail_with("");
                return false;
            }
            ms_wait += ms_since(t0);
            t0 = Clock::now();
            for (size_t r = 0; r < b.runs[d]; r++, run_count++) print_run(run_count, pk + r * GPSACQ_NUM_SATS);
            fflush(stdout);
            ms_print += ms_since(t0);
        }
        return true;
    };

    while (!eof && g_status == 0) {
        Batch b;
        b.slot = slot;
        b.runs.assign(n_dev, 0);
        size_t total = 0;
        for (size_t d = 0; d < n_dev && !eof; d++) {
            const size_t want = ramp * run_bytes;
            unsigned char *buf = gpsacq_pipe_buffer(g_engines[d], slot, max_runs * run_bytes);
            if (!buf) {
                fail_with("");
                break;
            }
            Clock::time_point t0 = Clock::now();
            size_t got;
            if (map) {
                struct stat st;  // the bytes the file holds NOW: a shorter file ends the search at its last whole run
                const size_t live = (fstat(fileno(fp), &st) == 0 && st.st_size >= 0 && (size_t)st.st_size < map_len) ? (size_t)st.st_size : map_len;
                const size_t avail = live > map_pos ? live - map_pos : 0;
                got = avail < want ? avail : want;
                copy_parallel(buf, map + map_pos, got);
                map_pos += got;
            } else {
                got = read_fully(fp, buf, want);
            }
            ms_read += ms_since(t0);
            if (got < want) eof = true;
            // a run is complete when all 32 of its Sample() calls got their 10 x 512 bytes (:135-140,239-244)
            const size_t runs = got / run_bytes;
            if (runs == 0) continue;
            t0 = Clock::now();
            iqin.first_sample = sample_pos;
            const int rc = gpsacq_pipe_submit(g_engines[d], slot, runs * GPSACQ_NUM_SATS, block_bytes, iq ? &iqin : NULL);
            ms_submit += ms_since(t0);
            if (rc != GPSACQ_OK) {  // a library function does not exit(): report, stop, leave the status for the caller
                fprintf(stderr, "gpsacq: %s\n", gpsacq_last_error());
                g_status = rc;
                break;
            }
            sample_pos += (uint64_t)runs * GPSACQ_NUM_SATS * GPSACQ_BLOCK_BYTES * 8;
            b.runs[d] = runs;
            total += runs;
        }
        if (total > 0) inflight.push_back(b);
        if (g_status != 0) break;
        // keep one batch in flight while the next is read, unless the file is finished
        while (inflight.size() > (eof ? 0u : 1u))
            if (!drain_one()) break;
        slot = (slot + 1) % GPSACQ_PIPE_SLOTS;
        if (ramp < max_runs) ramp = ramp * 2 < max_runs ? ramp * 2 : max_runs;
    }
    if (g_status != 0) {
        // finish what is in flight so that no search outlives the call, print nothing more
        while (!inflight.empty()) {
            Batch b = inflight.front();
            inflight.pop_front();
            for (size_t d = 0; d < n_dev; d++)
                if (b.runs[d]) (void)gpsacq_pipe_collect(g_engines[d], b.slot, NULL, NULL);
        }
    } else {
        printf("run out of file!\n");
    }
    fclose(fp);
    if (trace)
        fprintf(stderr,
                "gpsacq trace: SearchInit %.1f ms | SearchTask %.1f ms = mean pass %.1f + buffers %.1f + read %.1f + submit %.1f + wait for GPU %.1f + report %.1f "
                "(+ overlap); %d runs, %zu device(s), input %s\n",
                g_init_ms, ms_since(t_start), ms_sums, ms_setup, ms_read, ms_submit, ms_wait, ms_print, run_count, n_dev, iq ? fmt : "bits");
}
