/*
 * gpsacq_oracle.c -- CPU restatement of the gps_test acquisition path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under gnss-gps-sdr_amd/ (the product) may
 * include, link or call this file.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py load it, and only as the checker / the timed CPU port.
 *
 * What it restates (all citations relative to /root/reference):
 *   c/cacode.h:9-35            C/A Gold-code LFSR               -> oracle_ca_chips()
 *   c/search_offline.cpp:16-53 PRN -> (T1,T2) tap table         -> SATS[]
 *   c/search_offline.cpp:74-110  SearchInit (code replica + FFT) -> oracle_code_replica(), oracle_init()
 *   c/search_offline.cpp:121-165 Sample (unpack, XOR mix, FFT)   -> oracle_mix_block(), oracle_sample()
 *   c/search_offline.cpp:169-201 Correlate                       -> oracle_correlate()
 *   c/search_offline.cpp:219-292 SearchTask (run loop + report)  -> oracle_search_file()
 *
 * Third-party arithmetic: the reference calls FFTW3 single precision
 * (fftwf_plan_dft_1d / fftwf_execute, N=40000, forward sign -1, backward +1, no 1/N;
 * call sites search_offline.cpp:78,79,105,161,187).  FFTW is not vendored, has no pinned
 * version (system /usr/lib/libfftw3f.a, Makefile:4) and is absent from this image, so the
 * reference is UNBUILDABLE here.  The DFT below is our own mixed-radix Stockham
 * transform (radix 4/5/2) computing the same published definition
 *     X[k] = sum_n x[n] exp(-/+ 2 pi i n k / N), unnormalised.
 * Build with -DORACLE_REAL=double (default; the checker: transforms in double, values
 * rounded to float wherever the reference stores fftwf_complex) or -DORACLE_REAL=float
 * (the timed CPU port; float transform like the FFTW path).
 *
 * PIN STATUS: parity UNPINNED at the level of FFTW's float rounding (no FFTW in this image; `make -C oracle ref` builds the
 * reference itself where FFTW exists and tests/test_ref_binary.py then pins this file to it).  Pinned to (0) the code phase of
 * PRN 8 in all 12 runs (and at all 384 blocks) of gps_sig_tmp.bin as it follows from gps_sig_gen.m's own parameters
 * (tests/test_oracle.py), (a) the README known answer (PRN 8, Doppler 0), (b) an independent float64 numpy restatement
 * (tests/golden/make_golden.py), (c) gps_sig_gen.m restated bit for bit (tests/test_siggen.py).  NOT a pin: the transcript in
 * tests/golden/ref_known_answers.json (BASELINE.md section 2 / SURVEY.md section 8c) comes from the survey's MKL-shim build
 * of the reference sources -- a stand-in build; it is compared as a regression transcript and pins nothing by itself.  At the
 * level of FFTW's own float rounding (~1e-7) parity is unpinned (no FFTW here).
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#ifndef ORACLE_REAL
#define ORACLE_REAL double
#endif
typedef ORACLE_REAL real_t;

#define FFT_LEN 40000 /* c/gps_offline.h:15 */
#define NUM_SATS 32   /* c/gps_offline.h:16 */
#define CPS 1.023e6   /* c/gps_offline.h:30 */
#define BLOCK_BYTES 5120   /* 10 packets x 512 B, search_offline.cpp:129,135-136 */
#define BLOCK_SAMPLES 40960

typedef struct { float re, im; } cf32;
typedef struct { real_t re, im; } cplx;

/* c/search_offline.cpp:20-53 -- only T1,T2 are used */
static const int SATS[NUM_SATS][2] = {
    {2, 6}, {3, 7}, {4, 8}, {5, 9}, {1, 9}, {2, 10}, {1, 8}, {2, 9}, {3, 10}, {2, 3}, {3, 4},
    {5, 6}, {6, 7}, {7, 8}, {8, 9}, {9, 10}, {1, 4}, {2, 5}, {3, 6}, {4, 7}, {5, 8}, {6, 9},
    {1, 3}, {4, 6}, {5, 7}, {6, 8}, {7, 9}, {8, 10}, {1, 6}, {2, 7}, {3, 8}, {4, 9}};

/* ------------------------------------------------------------------------------------- */
/* c/cacode.h:9-35: G1 = x^10+x^3+1, G2 = x^10+x^9+x^8+x^6+x^3+x^2+1, all-ones start,    */
/* chip = g1[10] ^ g2[T1] ^ g2[T2].                                                       */
typedef struct { unsigned char g1[11], g2[11]; int t1, t2; } ca_t;
static void ca_init(ca_t *c, int t1, int t2) {
    memset(c, 0, sizeof *c);
    for (int i = 1; i <= 10; i++) c->g1[i] = c->g2[i] = 1;
    c->t1 = t1; c->t2 = t2;
}
static int ca_chip(const ca_t *c) { return c->g1[10] ^ c->g2[c->t1] ^ c->g2[c->t2]; }
static void ca_clock(ca_t *c) {
    c->g1[0] = c->g1[3] ^ c->g1[10];
    c->g2[0] = c->g2[2] ^ c->g2[3] ^ c->g2[6] ^ c->g2[8] ^ c->g2[9] ^ c->g2[10];
    memmove(c->g1 + 1, c->g1, 10);
    memmove(c->g2 + 1, c->g2, 10);
}
static unsigned ca_get_g1(const ca_t *c) { /* cacode.h:30-34 */
    unsigned ret = 0;
    for (int bit = 0; bit < 10; bit++) ret += ret + c->g1[10 - bit];
    return ret;
}

/* one period (1023 chips) of PRN sv (0-based), chips as 0/1 */
void oracle_ca_chips(int sv, unsigned char *chips) {
    ca_t c; ca_init(&c, SATS[sv][0], SATS[sv][1]);
    for (int i = 0; i < 1023; i++) { chips[i] = (unsigned char)ca_chip(&c); ca_clock(&c); }
}

/* c/search_offline.cpp:205-209 */
int oracle_search_code(int sv, int g1) {
    ca_t c; ca_init(&c, SATS[sv][0], SATS[sv][1]);
    int chips = 0;
    while (ca_get_g1(&c) != (unsigned)g1) { ca_clock(&c); chips++; if (chips > 2048) return -1; }
    return chips;
}

static inline float bipolar(int bit) { return bit ? -1.0f : 1.0f; } /* :68-70 */

/* c/search_offline.cpp:76,83-103: float ca_rate, float phase accumulator, comparisons and
 * the blend arithmetic promoted to double exactly as the C expressions are.               */
void oracle_code_replica(double fs, int sv, float *out /* FFT_LEN */) {
    const float ca_rate = (float)(CPS / fs);
    ca_t c; ca_init(&c, SATS[sv][0], SATS[sv][1]);
    float ca_phase = 0;
    for (int i = 0; i < FFT_LEN; i++) {
        float chip = bipolar(ca_chip(&c));
        ca_phase += ca_rate;
        if (ca_phase >= 1.0) {
            ca_phase -= 1.0;                 /* float = (double)float - 1.0 */
            ca_clock(&c);
            chip *= 1.0 - ca_phase;          /* float = (double)chip * (1.0 - (double)phase) */
            chip += ca_phase * bipolar(ca_chip(&c));
        }
        out[i] = chip;
    }
}

/* c/search_offline.cpp:124-127,131,155-156: quadrant index int(lo_phase) per sample,       */
/* lo_rate = float(4*FC/FS), float accumulator, wrap at >= 4.  n may exceed FFT_LEN (the    */
/* reference keeps accumulating over all 40960 samples it reads).                           */
void oracle_lo_quadrants(double fc, double fs, int n, unsigned char *quad) {
    const float lo_rate = (float)(4 * fc / fs);
    float lo_phase = 0;
    for (int i = 0; i < n; i++) {
        quad[i] = (unsigned char)(int)lo_phase;
        lo_phase += lo_rate;
        if (lo_phase >= 4) lo_phase -= 4;
    }
}

/* c/search_offline.cpp:141-153: LSB-first unpack, I = Bipolar(bit^lo_cos[q]),              */
/* Q = Bipolar(bit^lo_sin[q]), lo_sin={1,1,0,0}, lo_cos={0,1,1,0}.  Produces all 40960       */
/* samples of the 5120-byte block (the reference writes all of them, the last 960 past the  */
/* end of fwd_buf -- SURVEY.md fact 5).                                                     */
void oracle_mix_block(const unsigned char *bytes, const unsigned char *quad, cf32 *out) {
    static const int lo_sin[4] = {1, 1, 0, 0}, lo_cos[4] = {0, 1, 1, 0};
    for (int i = 0; i < BLOCK_SAMPLES; i++) {
        int bit = (bytes[i >> 3] >> (i & 7)) & 1;
        out[i].re = bipolar(bit ^ lo_cos[quad[i]]);
        out[i].im = bipolar(bit ^ lo_sin[quad[i]]);
    }
}

/* ------------------------------------------------------------------------------------- */
/* Mixed-radix Stockham DFT (own code).  n = prod radix[i], radix in {2,4,5}.             */
typedef struct {
    int n, nstage, radix[24];
    cplx *w;      /* w[k] = exp(-2 pi i k / n), computed in long double */
    cplx *a, *b;  /* ping-pong work buffers */
} fft_plan;

static fft_plan *fft_plan_create(int n) {
    fft_plan *p = (fft_plan *)calloc(1, sizeof *p);
    p->n = n;
    int m = n;
    while (m % 4 == 0) { p->radix[p->nstage++] = 4; m /= 4; }
    while (m % 2 == 0) { p->radix[p->nstage++] = 2; m /= 2; }
    while (m % 5 == 0) { p->radix[p->nstage++] = 5; m /= 5; }
    if (m != 1) { free(p); return NULL; }
    p->w = (cplx *)malloc(sizeof(cplx) * n);
    p->a = (cplx *)malloc(sizeof(cplx) * n);
    p->b = (cplx *)malloc(sizeof(cplx) * n);
    const long double tp = 6.283185307179586476925286766559005768L;
    for (int k = 0; k < n; k++) {
        long double th = tp * (long double)k / (long double)n;
        p->w[k].re = (real_t)cosl(th);
        p->w[k].im = (real_t)(-sinl(th));
    }
    return p;
}
static void fft_plan_destroy(fft_plan *p) { if (p) { free(p->w); free(p->a); free(p->b); free(p); } }

static inline cplx cmul(cplx a, cplx b) { cplx r = {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; return r; }
static inline cplx cadd(cplx a, cplx b) { cplx r = {a.re + b.re, a.im + b.im}; return r; }
static inline cplx csub(cplx a, cplx b) { cplx r = {a.re - b.re, a.im - b.im}; return r; }
/* multiply by -i*sg (sg=+1 forward, -1 inverse) */
static inline cplx cmul_mi(cplx a, int sg) { cplx r; if (sg > 0) { r.re = a.im; r.im = -a.re; } else { r.re = -a.im; r.im = a.re; } return r; }

/* One Stockham DIF pass: sequence length nn (current), stride s, radix r.
 * y[q + s*(r*p + k)] = (sum_i x[q + s*(p + i*m)] W_r^{ik}) * W_nn^{pk},  m = nn/r.  */
static void stockham_pass(const fft_plan *pl, int nn, int s, int r, int sg, const cplx *x, cplx *y) {
    const int m = nn / r, n = pl->n, tstep = n / nn;
    const real_t C1 = (real_t)0.30901699437494742410L, C2 = (real_t)-0.80901699437494742410L;
    const real_t S1 = (real_t)0.95105651629515357212L, S2 = (real_t)0.58778525229247312917L;
    for (int p = 0; p < m; p++) {
        cplx w1 = pl->w[(size_t)p * tstep % n];          /* exact table values, not products: */
        cplx w2 = pl->w[(size_t)2 * p * tstep % n];      /* keeps twiddle error at rounding level */
        cplx w3 = pl->w[(size_t)3 * p * tstep % n];
        cplx w4 = pl->w[(size_t)4 * p * tstep % n];
        if (sg < 0) { w1.im = -w1.im; w2.im = -w2.im; w3.im = -w3.im; w4.im = -w4.im; }
        const cplx *xp = x + (size_t)s * p;
        cplx *yp = y + (size_t)s * r * p;
        if (r == 2) {
            for (int q = 0; q < s; q++) {
                cplx a = xp[q], b = xp[q + (size_t)s * m];
                yp[q] = cadd(a, b);
                yp[q + s] = cmul(csub(a, b), w1);
            }
        } else if (r == 4) {
            for (int q = 0; q < s; q++) {
                cplx a = xp[q], b = xp[q + (size_t)s * m], c = xp[q + (size_t)s * 2 * m], d = xp[q + (size_t)s * 3 * m];
                cplx apc = cadd(a, c), amc = csub(a, c), bpd = cadd(b, d), bmd = cmul_mi(csub(b, d), sg);
                yp[q] = cadd(apc, bpd);
                yp[q + s] = cmul(cadd(amc, bmd), w1);
                yp[q + 2 * s] = cmul(csub(apc, bpd), w2);
                yp[q + 3 * s] = cmul(csub(amc, bmd), w3);
            }
        } else { /* r == 5 */
            for (int q = 0; q < s; q++) {
                cplx x0 = xp[q], x1 = xp[q + (size_t)s * m], x2 = xp[q + (size_t)s * 2 * m],
                     x3 = xp[q + (size_t)s * 3 * m], x4 = xp[q + (size_t)s * 4 * m];
                cplx t1 = cadd(x1, x4), t2 = cadd(x2, x3), t3 = csub(x1, x4), t4 = csub(x2, x3);
                cplx m1 = {x0.re + C1 * t1.re + C2 * t2.re, x0.im + C1 * t1.im + C2 * t2.im};
                cplx m2 = {x0.re + C2 * t1.re + C1 * t2.re, x0.im + C2 * t1.im + C1 * t2.im};
                cplx s1 = {S1 * t3.re + S2 * t4.re, S1 * t3.im + S2 * t4.im};
                cplx s2 = {S2 * t3.re - S1 * t4.re, S2 * t3.im - S1 * t4.im};
                cplx is1 = cmul_mi(s1, sg), is2 = cmul_mi(s2, sg); /* -i*s forward */
                yp[q] = cadd(x0, cadd(t1, t2));
                yp[q + s] = cmul(cadd(m1, is1), w1);
                yp[q + 2 * s] = cmul(cadd(m2, is2), w2);
                yp[q + 3 * s] = cmul(csub(m2, is2), w3);
                yp[q + 4 * s] = cmul(csub(m1, is1), w4);
            }
        }
    }
}

/* in-place (on cf32 storage) unnormalised DFT; sg=+1 forward (exp(-i..)), -1 backward */
static void fft_exec_cf32(fft_plan *pl, cf32 *buf, int sg) {
    const int n = pl->n;
    cplx *x = pl->a, *y = pl->b;
    for (int i = 0; i < n; i++) { x[i].re = buf[i].re; x[i].im = buf[i].im; }
    int nn = n, s = 1;
    for (int st = 0; st < pl->nstage; st++) {
        int r = pl->radix[st];
        stockham_pass(pl, nn, s, r, sg, x, y);
        cplx *t = x; x = y; y = t;
        nn /= r; s *= r;
    }
    for (int i = 0; i < n; i++) { buf[i].re = (float)x[i].re; buf[i].im = (float)x[i].im; }
}

/* exported for tests: DFT of n complex floats (interleaved), dir=-1 forward, +1 backward (FFTW signs) */
int oracle_dft(int n, int dir, const float *in, float *out) {
    fft_plan *pl = fft_plan_create(n);
    if (!pl) return -1;
    cf32 *buf = (cf32 *)malloc(sizeof(cf32) * n);
    memcpy(buf, in, sizeof(cf32) * n);
    fft_exec_cf32(pl, buf, dir < 0 ? +1 : -1);
    memcpy(out, buf, sizeof(cf32) * n);
    free(buf); fft_plan_destroy(pl);
    return 0;
}

/* ------------------------------------------------------------------------------------- */
typedef struct {
    double fc, fs, max_fo;
    int dmax, nlags;        /* Doppler half-range in bins (:176), lags scanned (:190) */
    int ref_quirks;         /* 1: emulate the fwd_buf overrun clobbering code[0][0..959] */
    fft_plan *plan;
    cf32 *code;             /* [NUM_SATS][FFT_LEN] code spectra (:105-106) */
    unsigned char *quad;    /* [BLOCK_SAMPLES] LO quadrants */
    cf32 *fwd;              /* [BLOCK_SAMPLES] fwd_buf (+ the 960 overrun samples) */
    cf32 *rev;              /* [FFT_LEN] rev_buf */
    cf32 *code0_save;       /* pristine code[0][0..959] when ref_quirks */
} oracle_t;

typedef struct { float max_pwr; int32_t max_i; float tot_pwr; float snr; } oracle_cell;
typedef struct { float snr; int32_t lo_shift; int32_t ca_shift; float max_pwr; } oracle_peak;

int oracle_dmax(double fs, double max_fo) { /* :176 the int conversion truncates toward zero */
    return (int)(max_fo * (double)FFT_LEN / (double)fs);
}
int oracle_nlags(double fs) { /* :190  for(i=0; i<FS/1000; i++) */
    int i = 0; while ((double)i < fs / 1000) i++; return i;
}

oracle_t *oracle_create(double fc, double fs, double max_fo, int ref_quirks) {
    oracle_t *o = (oracle_t *)calloc(1, sizeof *o);
    o->fc = fc; o->fs = fs; o->max_fo = max_fo; o->ref_quirks = ref_quirks;
    o->dmax = oracle_dmax(fs, max_fo);
    o->nlags = oracle_nlags(fs);
    if (o->nlags > FFT_LEN) o->nlags = FFT_LEN;
    o->plan = fft_plan_create(FFT_LEN);
    o->code = (cf32 *)malloc(sizeof(cf32) * NUM_SATS * FFT_LEN);
    o->quad = (unsigned char *)malloc(BLOCK_SAMPLES);
    o->fwd = (cf32 *)malloc(sizeof(cf32) * BLOCK_SAMPLES);
    o->rev = (cf32 *)malloc(sizeof(cf32) * FFT_LEN);
    o->code0_save = (cf32 *)malloc(sizeof(cf32) * 960);
    oracle_lo_quadrants(fc, fs, BLOCK_SAMPLES, o->quad);
    float *rep = (float *)malloc(sizeof(float) * FFT_LEN);
    for (int sv = 0; sv < NUM_SATS; sv++) { /* SearchInit :81-107 */
        oracle_code_replica(fs, sv, rep);
        cf32 *c = o->code + (size_t)sv * FFT_LEN;
        for (int i = 0; i < FFT_LEN; i++) { c[i].re = rep[i]; c[i].im = 0; }
        fft_exec_cf32(o->plan, c, +1);
    }
    memcpy(o->code0_save, o->code, sizeof(cf32) * 960);
    free(rep);
    return o;
}
void oracle_destroy(oracle_t *o) {
    if (!o) return;
    fft_plan_destroy(o->plan);
    free(o->code); free(o->quad); free(o->fwd); free(o->rev); free(o->code0_save); free(o);
}
int oracle_get_dmax(const oracle_t *o) { return o->dmax; }
int oracle_get_nlags(const oracle_t *o) { return o->nlags; }
void oracle_get_code_spectrum(const oracle_t *o, int sv, float *out) {
    memcpy(out, o->code + (size_t)sv * FFT_LEN, sizeof(cf32) * FFT_LEN);
    if (sv == 0) memcpy(out, o->code0_save, sizeof(cf32) * 960);
}

/* Sample(): :121-165 on one 5120-byte block already in memory. */
void oracle_sample(oracle_t *o, const unsigned char *bytes) {
    oracle_mix_block(bytes, o->quad, o->fwd);
    if (o->ref_quirks) /* BSS layout rev_buf|fwd_buf|code: samples 40000.. land on code[0][0..959] */
        memcpy(o->code, o->fwd + FFT_LEN, sizeof(cf32) * 960);
    fft_exec_cf32(o->plan, o->fwd, +1); /* :161 */
}
void oracle_get_sample_spectrum(const oracle_t *o, float *out) { memcpy(out, o->fwd, sizeof(cf32) * FFT_LEN); }

/* EXTENSION, no reference behaviour ("arbitrary Doppler step"): Sample() whose mixed
 * samples are multiplied by exp(-2 pi i eps n / N) before the transform -- a carrier offset of eps FFT bins
 * (0 <= eps < 1), so that Correlate's whole-bin shift d (:182) then tests the Doppler (d + eps) fs/N.  The ramp is
 * formed in double and the product rounded to float (what a float sample buffer would hold); eps = 0 is
 * oracle_sample().  Restates the product's sub-bin grid (gpsacq_set_doppler_step). */
void oracle_sample_ramped(oracle_t *o, const unsigned char *bytes, double eps) {
    oracle_mix_block(bytes, o->quad, o->fwd);
    if (eps != 0.0) {
        const double tp = 6.283185307179586476925286766559;
        for (int n = 0; n < FFT_LEN; n++) {
            const double th = tp * eps * (double)n / (double)FFT_LEN, c = cos(th), s = sin(th);
            const double re = o->fwd[n].re, im = o->fwd[n].im;
            o->fwd[n].re = (float)(re * c + im * s);
            o->fwd[n].im = (float)(im * c - re * s);
        }
    }
    fft_exec_cf32(o->plan, o->fwd, +1);
}

/* Correlate(): :169-201.  cells (may be NULL) receives one record per Doppler bin. */
float oracle_correlate(oracle_t *o, int sv, int *max_snr_dop, int *max_snr_i, oracle_cell *cells) {
    const cf32 *data = o->fwd;
    const cf32 *code = o->code + (size_t)sv * FFT_LEN;
    cf32 *prod = o->rev;
    float max_snr = 0;
    const int dmax = o->dmax, S = o->nlags;
    for (int dop = -dmax; dop <= dmax; dop++) {
        float max_pwr = 0, tot_pwr = 0;
        int max_pwr_i = 0; /* reference leaves it uninitialised; never read unless pwr>0 occurred */
        for (int i = 0; i < FFT_LEN; i++) { /* :181-185, float products, left-to-right */
            int j = (i - dop + FFT_LEN) % FFT_LEN;
            if (dop > FFT_LEN || dop < -FFT_LEN) j = ((i - dop) % FFT_LEN + FFT_LEN) % FFT_LEN;
            prod[i].re = data[i].re * code[j].re + data[i].im * code[j].im;
            prod[i].im = data[i].re * code[j].im - data[i].im * code[j].re;
        }
        fft_exec_cf32(o->plan, prod, -1); /* :187 backward, unnormalised */
        int i;
        for (i = 0; i < S; i++) { /* :190-194 */
            float pwr = prod[i].re * prod[i].re + prod[i].im * prod[i].im;
            if (pwr > max_pwr) max_pwr = pwr, max_pwr_i = i;
            tot_pwr += pwr;
        }
        float ave_pwr = tot_pwr / i;
        float snr = max_pwr / ave_pwr;
        if (!(tot_pwr > 0)) snr = 0; /* build-defined: the reference would compute 0/0 */
        if (cells) { cells[dop + dmax].max_pwr = max_pwr; cells[dop + dmax].max_i = max_pwr_i;
                     cells[dop + dmax].tot_pwr = tot_pwr; cells[dop + dmax].snr = snr; }
        if (snr > max_snr) max_snr = snr, *max_snr_dop = dop, *max_snr_i = max_pwr_i;
    }
    return max_snr;
}

/* |IFFT|^2 over the scanned lags for one Doppler bin of the block last given to oracle_sample()
 * (the inner part of Correlate, :181-191).  Used to restate the non-coherent extension:
 * the caller sums these arrays over blocks and scans the sum like :190-196. */
void oracle_cell_power(oracle_t *o, int sv, int dop, float *pwr /* nlags */) {
    const cf32 *data = o->fwd;
    const cf32 *code = o->code + (size_t)sv * FFT_LEN;
    cf32 *prod = o->rev;
    for (int i = 0; i < FFT_LEN; i++) {
        int j = ((i - dop) % FFT_LEN + FFT_LEN) % FFT_LEN;
        prod[i].re = data[i].re * code[j].re + data[i].im * code[j].im;
        prod[i].im = data[i].re * code[j].im - data[i].im * code[j].re;
    }
    fft_exec_cf32(o->plan, prod, -1);
    for (int i = 0; i < o->nlags; i++) pwr[i] = prod[i].re * prod[i].re + prod[i].im * prod[i].im;
}

/* One cell of Correlate (:178-196) at whole-bin shift dop of the block last sampled. */
void oracle_one_cell(oracle_t *o, int sv, int dop, oracle_cell *out) {
    const cf32 *data = o->fwd;
    const cf32 *code = o->code + (size_t)sv * FFT_LEN;
    cf32 *prod = o->rev;
    const int S = o->nlags;
    for (int i = 0; i < FFT_LEN; i++) {
        int j = ((i - dop) % FFT_LEN + FFT_LEN) % FFT_LEN;
        prod[i].re = data[i].re * code[j].re + data[i].im * code[j].im;
        prod[i].im = data[i].re * code[j].im - data[i].im * code[j].re;
    }
    fft_exec_cf32(o->plan, prod, -1);
    float max_pwr = 0, tot_pwr = 0;
    int max_pwr_i = 0, i;
    for (i = 0; i < S; i++) {
        float pwr = prod[i].re * prod[i].re + prod[i].im * prod[i].im;
        if (pwr > max_pwr) max_pwr = pwr, max_pwr_i = i;
        tot_pwr += pwr;
    }
    out->max_pwr = max_pwr; out->max_i = max_pwr_i; out->tot_pwr = tot_pwr;
    out->snr = (tot_pwr > 0) ? max_pwr / (tot_pwr / i) : 0;
}

/* One (block, sv) search = Sample + Correlate */
void oracle_search_block(oracle_t *o, const unsigned char *bytes, int sv, oracle_cell *cells, oracle_peak *peak) {
    int lo = 0, ca = 0;
    oracle_sample(o, bytes);
    float snr = oracle_correlate(o, sv, &lo, &ca, cells);
    if (peak) {
        peak->snr = snr; peak->lo_shift = lo; peak->ca_shift = ca; peak->max_pwr = 0;
        if (cells) peak->max_pwr = cells[lo + o->dmax].max_pwr;
    }
}

/* SearchTask(): :219-292.  Writes the report (without the 6 banner lines) into out (size cap);
 * max_runs<=0: until the file runs out.  Also returns per-(run,sv) peaks when peaks!=NULL
 * (capacity peaks_cap records).  Returns number of complete runs, or -1 if the file cannot
 * be opened (after writing "can not open file!\n").                                        */
int oracle_search_file(oracle_t *o, const char *path, int max_runs, char *out, size_t cap,
                       oracle_peak *peaks, size_t peaks_cap) {
    size_t len = 0;
#define EMIT(...) do { if (len < cap) { int k_ = snprintf(out + len, cap - len, __VA_ARGS__); if (k_ > 0) len += (size_t)k_; } } while (0)
    FILE *fp = fopen(path, "rb");
    if (!fp) { EMIT("can not open file!\n"); return -1; }
    unsigned char bytes[BLOCK_BYTES];
    int run_out = 0, run_count = 0;
    float sat_snr_store[NUM_SATS], snr_store[NUM_SATS];
    int sv_store[NUM_SATS], lo_store[NUM_SATS], ca_store[NUM_SATS];
    for (;;) {
        int hit_count = 0;
        if (max_runs > 0 && run_count >= max_runs) break;
        for (int sv = 0; sv < NUM_SATS; sv++) {
            size_t got = 0;
            for (int pk = 0; pk < 10; pk++) { /* 10 x fread(512) :135-140 */
                size_t r = fread(bytes + 512 * pk, 1, 512, fp);
                if (r != 512) { run_out = 1; break; }
                got += r;
            }
            if (run_out) { EMIT("run out of file!\n"); break; }
            oracle_peak pk;
            oracle_search_block(o, bytes, sv, NULL, &pk);
            sat_snr_store[sv] = pk.snr;
            if (peaks && (size_t)(run_count * NUM_SATS + sv) < peaks_cap) peaks[run_count * NUM_SATS + sv] = pk;
            if (pk.snr < 25) continue;
            snr_store[hit_count] = pk.snr; sv_store[hit_count] = sv;
            lo_store[hit_count] = pk.lo_shift; ca_store[hit_count] = pk.ca_shift; hit_count++;
        }
        if (run_out) break;
        EMIT("%2d satellite: ", run_count); for (int i = 0; i < hit_count; i++) EMIT("%5d ", sv_store[i]); EMIT("\n");
        EMIT("%2d SNR(>=25): ", run_count); for (int i = 0; i < hit_count; i++) EMIT("%5.1f ", snr_store[i]); EMIT("\n");
        EMIT("%2d  lo_shift: ", run_count); for (int i = 0; i < hit_count; i++) EMIT("%5d ", lo_store[i]); EMIT("\n");
        EMIT("%2d  ca_shift: ", run_count); for (int i = 0; i < hit_count; i++) EMIT("%5d ", ca_store[i]); EMIT("\n");
        for (int sv = 0; sv < NUM_SATS; sv++) EMIT("%2.0f ", sat_snr_store[sv]);
        EMIT("\n\n");
        run_count++;
    }
    fclose(fp);
#undef EMIT
    return run_count;
}

/* ------------------------------------------------------------------------------------- */
/* Timed CPU port for bench.py's cpu_baseline: search n_blocks blocks (block b against PRN */
/* b%32, like SearchTask), return number of (PRN,Doppler) cells processed.                  */
long oracle_bench_blocks(oracle_t *o, const unsigned char *bits, long n_blocks, long stride, oracle_peak *peaks) {
    long cells = 0;
    for (long b = 0; b < n_blocks; b++) {
        oracle_peak pk;
        oracle_search_block(o, bits + b * stride, (int)(b % NUM_SATS), NULL, &pk);
        if (peaks) peaks[b] = pk;
        cells += 2 * o->dmax + 1;
    }
    return cells;
}

/* The same port on several cores for bench.py's all-cores figure: one oracle instance per OpenMP thread (SearchInit of every
 * instance untimed, behind a barrier), then the threads take blocks of the sample round-robin (block b against PRN b % 32, the
 * reference schedule) until `seconds` have passed.  Returns the cells searched; *elapsed gets the timed span and
 * *threads_used what the runtime really gave.  Test infrastructure, like the rest of this file. */
#ifdef _OPENMP
#include <omp.h>
long oracle_bench_omp(double fc, double fs, double max_fo, const unsigned char *bits, long n_blocks, long stride, int nthreads,
                      double seconds, double *elapsed, int *threads_used) {
    long cells = 0;
    double t_begin = 0, t_end = 0;
    int used = 0;
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel num_threads(nthreads) reduction(+ : cells)
    {
        oracle_t *o = oracle_create(fc, fs, max_fo, 0);
#pragma omp barrier
#pragma omp master
        {
            t_begin = omp_get_wtime();
            used = omp_get_num_threads();
        }
#pragma omp barrier
        const int tid = omp_get_thread_num(), nt = omp_get_num_threads();
        const double deadline = omp_get_wtime() + seconds;
        for (long b = tid; o && omp_get_wtime() < deadline; b += nt) {
            oracle_peak pk;
            const long blk = b % n_blocks;
            oracle_search_block(o, bits + blk * stride, (int)(blk % NUM_SATS), NULL, &pk);
            cells += 2 * o->dmax + 1;
        }
#pragma omp barrier
#pragma omp master
        t_end = omp_get_wtime();
        if (o) oracle_destroy(o);
    }
    if (elapsed) *elapsed = t_end - t_begin;
    if (threads_used) *threads_used = used;
    return cells;
}

/* ONE pass over a capture on several cores that KEEPS what it computed (bench.py: the all-cores figure and, with the same work, the
 * checker of every block of the timed step): block b (capture index first_block + b) against PRN (first_block + b) % 32, the
 * reference schedule (SearchTask :239-246); blocks are handed out in ascending order, a thread takes no new block after `seconds`.
 * peaks[b] / cells[b * ndop ..] (cells may be NULL) / done[b] = 1 are written for every block searched; blocks never reached keep
 * done[b] = 0.  Returns the cells searched; *elapsed the timed span (SearchInit of every instance untimed, like oracle_bench_omp). */
long oracle_search_omp(double fc, double fs, double max_fo, const unsigned char *bits, long n_blocks, long stride, long first_block,
                       int nthreads, double seconds, oracle_peak *peaks, oracle_cell *cells, unsigned char *done, double *elapsed,
                       int *threads_used) {
    long total = 0, next = 0;
    double t_begin = 0, t_end = 0;
    int used = 0;
    if (nthreads < 1) nthreads = 1;
    if (done) memset(done, 0, (size_t)n_blocks);
#pragma omp parallel num_threads(nthreads) reduction(+ : total)
    {
        oracle_t *o = oracle_create(fc, fs, max_fo, 0);
#pragma omp barrier
#pragma omp master
        {
            t_begin = omp_get_wtime();
            used = omp_get_num_threads();
        }
#pragma omp barrier
        const double deadline = omp_get_wtime() + seconds;
        const int ndop = o ? 2 * o->dmax + 1 : 0;
        while (o && omp_get_wtime() < deadline) {
            long b;
#pragma omp atomic capture
            b = next++;
            if (b >= n_blocks) break;
            oracle_peak pk;
            oracle_cell *row = cells ? cells + (size_t)b * ndop : NULL;
            oracle_search_block(o, bits + b * stride, (int)((first_block + b) % NUM_SATS), row, &pk);
            if (!row) pk.max_pwr = 0;
            if (peaks) peaks[b] = pk;
            if (done) done[b] = 1;
            total += ndop;
        }
#pragma omp barrier
#pragma omp master
        t_end = omp_get_wtime();
        if (o) oracle_destroy(o);
    }
    if (elapsed) *elapsed = t_end - t_begin;
    if (threads_used) *threads_used = used;
    return total;
}
#endif

