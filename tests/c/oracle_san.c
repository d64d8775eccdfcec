/* oracle_san.c -- the CPU oracle under AddressSanitizer + UndefinedBehaviorSanitizer (the reference
 * itself overruns fwd_buf by 960 samples, c/search_offline.cpp:135-157; the restatement emulates that explicitly and must
 * stay clean).  Includes the oracle's one source file so that every static helper is instrumented too.
 * Usage: oracle_san <capture> <fc> <fs>; walks every entry point once (both quirk modes, a short file, a missing file, the
 * sub-bin ramp, the odd-size DFT); exit 0 when nothing fired (the sanitizers abort otherwise). */
#include "../../oracle/gpsacq_oracle.c"

int main(int argc, char **argv) {
    if (argc < 4) return 64;
    const double fc = atof(argv[2]), fs = atof(argv[3]);
    static unsigned char bits[3 * BLOCK_BYTES];
    FILE *fp = fopen(argv[1], "rb");
    if (!fp || fread(bits, 1, sizeof bits, fp) != sizeof bits) return 66;
    fclose(fp);
    int hits = 0;
    for (int quirks = 0; quirks < 2; quirks++) {
        oracle_t *o = oracle_create(fc, fs, 5000.0, quirks);
        const int nd = 2 * oracle_get_dmax(o) + 1;
        oracle_cell *cells = (oracle_cell *)malloc(sizeof(oracle_cell) * (size_t)nd);
        oracle_peak pk;
        oracle_search_block(o, bits, 0, cells, &pk);            /* PRN index 0: the one the overrun touches */
        oracle_search_block(o, bits + BLOCK_BYTES, 7, cells, &pk);
        hits += pk.snr >= 25;
        oracle_search_block(o, bits + 2 * BLOCK_BYTES, 31, NULL, &pk);
        float *pwr = (float *)malloc(sizeof(float) * (size_t)oracle_get_nlags(o));
        oracle_cell_power(o, 7, -oracle_get_dmax(o), pwr);
        oracle_cell_power(o, 7, oracle_get_dmax(o), pwr);
        oracle_sample_ramped(o, bits, 0.5);
        oracle_one_cell(o, 7, 0, cells);
        float *spec = (float *)malloc(sizeof(float) * 2 * FFT_LEN);
        oracle_get_code_spectrum(o, 0, spec);
        oracle_get_sample_spectrum(o, spec);
        char report[4096];
        oracle_peak peaks[NUM_SATS];
        /* a file of 3 blocks: "run out of file!" inside the first run; and a path that does not exist */
        if (oracle_search_file(o, argv[1], 1, report, sizeof report, peaks, NUM_SATS) < 0) return 70;
        if (oracle_search_file(o, "/nonexistent/capture.bin", 1, report, 8, NULL, 0) != -1) return 71; /* cap smaller than the message */
        free(spec); free(pwr); free(cells);
        oracle_destroy(o);
    }
    {   /* the DFT on sizes with every radix and a prime the plan must refuse or handle */
        static const int sizes[] = {1, 2, 4, 5, 8, 20, 25, 100, 1000};
        for (unsigned k = 0; k < sizeof sizes / sizeof *sizes; k++) {
            const int n = sizes[k];
            float *in = (float *)calloc((size_t)n * 2, sizeof(float)), *out = (float *)calloc((size_t)n * 2, sizeof(float));
            in[0] = 1.f;
            oracle_dft(n, +1, in, out);
            oracle_dft(n, -1, in, out);
            free(in); free(out);
        }
        float in7[14] = {1}, out7[14];
        (void)oracle_dft(7, +1, in7, out7);
    }
#ifdef _OPENMP
    {   /* the all-cores port of bench.py's cpu_baseline, two threads for a moment */
        double elapsed = 0;
        int used = 0;
        if (oracle_bench_omp(fc, fs, 5000.0, bits, 3, BLOCK_BYTES, 2, 0.2, &elapsed, &used) <= 0 || used < 1) return 73;
        /* the one-pass variant that keeps its results (the checker of bench.py's parity verdict): all three blocks, then a pass
         * whose time is up before it starts */
        const int nd = 2 * oracle_dmax(fs, 5000.0) + 1;
        oracle_cell *cells = (oracle_cell *)malloc(sizeof(oracle_cell) * 3 * (size_t)nd);
        oracle_peak pk3[3];
        unsigned char done[3];
        if (oracle_search_omp(fc, fs, 5000.0, bits, 3, BLOCK_BYTES, 32, 2, 30.0, pk3, cells, done, &elapsed, &used) != 3L * nd) return 74;
        if (!(done[0] && done[1] && done[2])) return 75;
        if (oracle_search_omp(fc, fs, 5000.0, bits, 3, BLOCK_BYTES, 0, 2, 0.0, pk3, NULL, done, &elapsed, &used) != 0 || done[0] || done[2]) return 76;
        free(cells);
    }
#endif
    unsigned char chips[1023];
    for (int sv = 0; sv < NUM_SATS; sv++) oracle_ca_chips(sv, chips);
    (void)oracle_search_code(0, 0x3ff);
    printf("oracle_san: ok (%d hits)\n", hits);
    return hits == 2 ? 0 : 72;
}
