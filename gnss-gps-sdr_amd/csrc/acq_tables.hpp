// acq_tables.hpp -- host-side construction of the constant tables the kernels read:
// twiddles (computed in long double, stored as float), the C/A code replicas
// (c/search_offline.cpp:74-103, c/cacode.h:9-35) and the quadrature-LO bit masks
// (c/search_offline.cpp:124-127,155-156).  Host C++ only.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "acq_phases.hpp"

namespace acq {

inline cf unit_fwd(long long num, long long den) {  // exp(-2 pi i num/den)
    num %= den;
    if (num < 0) num += den;
    const long double tp = 6.283185307179586476925286766559005768L;
    long double th = tp * (long double)num / (long double)den;
    return mk((float)cosl(th), (float)(-sinl(th)));
}

struct Tables {
    std::vector<cf> t1;  // [10][500]   W_5000^{j' alpha}
    std::vector<cf> t2;  // [25][20]    W_500^{j'' beta}
    std::vector<cf> bq;  // [8][250]    W_40000^{q rho}, indexed by the radix-20 butterfly rho = 10 beta + alpha a pass-3 thread owns
    std::vector<cf> wq;  // [8][40]     W_160^{q m}
    std::vector<cf> tn;  // [8][5000]   W_40000^{n' kappa}: forward transform's decimation-in-frequency twiddle
    Tables() : t1(RA * NBF1), t2((size_t)NT2), bq((size_t)NPOLY * NBF3), wq(NPOLY * WQ_STRIDE), tn((size_t)NPOLY * M_SUB) {
        for (int ka = 0; ka < NPOLY; ++ka)
            for (int n = 0; n < M_SUB; ++n) tn[(size_t)ka * M_SUB + n] = unit_fwd((long long)n * ka, N_FFT);
        for (int al = 0; al < RA; ++al)
            for (int jp = 0; jp < NBF1; ++jp) t1[al * NBF1 + jp] = unit_fwd((long long)jp * al, M_SUB);
        for (int be = 0; be < RB; ++be)
            for (int jpp = 0; jpp < RC; ++jpp) t2[(size_t)be * RC + jpp] = unit_fwd((long long)jpp * be, NBF1);
        for (int q = 0; q < NPOLY; ++q)
            for (int rho = 0; rho < NBF3; ++rho) bq[(size_t)q * NBF3 + rho] = unit_fwd((long long)q * rho, N_FFT);
        for (int q = 0; q < NPOLY; ++q)
            for (int m = 0; m < WQ_STRIDE; ++m) wq[q * WQ_STRIDE + m] = unit_fwd((long long)q * m, NW160);
    }
};

// Tables of the folded radix-8 rotation (k_corr<..., FOLD>): the rotation W_40000^{q n} of sub-transform q's outputs,
// n = 250 m + 10 beta + alpha, split as  W_4000^{q beta} (into pass 2's output twiddle: one 500-entry table per q)  x
// W_40000^{q (250 m + alpha)} (the accumulate's factor: per q a [10][160] table a workgroup copies its columns of into LDS),
// so that pass 3 multiplies nothing by a per-thread factor any more (20 complex multiplies per thread and sub-transform).
struct TablesFold {
    std::vector<cf> t2q;  // [8][25][20]   W_500^{j'' beta} W_4000^{q beta}
    std::vector<cf> tq;   // [8][10][160]  W_40000^{q (250 m + alpha)}
    TablesFold() : t2q((size_t)NPOLY * NT2), tq((size_t)NPOLY * RA * NW160 + 8, mk(0.f, 0.f)) {  // (+ tail padding, see k_corr)
        for (int q = 0; q < NPOLY; ++q) {
            for (int be = 0; be < RB; ++be)
                for (int jpp = 0; jpp < RC; ++jpp)  // j'' beta / 500 + q beta / 4000 = (8 j'' + q) beta / 4000
                    t2q[((size_t)q * RB + be) * RC + jpp] = unit_fwd((long long)(8 * jpp + q) * be, 4000);
            for (int al = 0; al < RA; ++al)
                for (int m = 0; m < NW160; ++m) tq[((size_t)q * RA + al) * NW160 + m] = unit_fwd((long long)q * (NBF3 * m + al), N_FFT);
        }
        fold.assign((size_t)NPOLY * FOLD_Q + 8, mk(0.f, 0.f));  // the device image: per q [t2q][tq], + tail padding
        for (int q = 0; q < NPOLY; ++q) {
            for (int i = 0; i < NT2; ++i) fold[(size_t)q * FOLD_Q + i] = t2q[(size_t)q * NT2 + i];
            for (int i = 0; i < RA * NW160; ++i) fold[(size_t)q * FOLD_Q + NT2 + i] = tq[(size_t)q * RA * NW160 + i];
        }
    }
    std::vector<cf> fold;
};

// Forward-transform tables for `sub` sub-bin Doppler offsets r/sub (r < sub) of a bin (extension; the reference's grid is
// sub = 1): spectrum r of a block is the transform of x[n] exp(-2 pi i (r/sub) n / N), i.e. the block's spectrum
// evaluated r/sub of a bin higher, which the decimation-in-frequency split absorbs exactly into its twiddles:
//   tn [r][kappa][n'] = exp(-2 pi i n' (kappa + r/sub) / N)
//   lutc[r][kappa][b]  = conj( sum_nu (+1 / -1 by bit nu of b: Bipolar(), :68-70) exp(-2 pi i nu (kappa + r/sub) / 8) ): the pruned radix-8 sum of the
//                        eight 1-bit samples a transposed byte holds, conjugated (k_fwd2 runs the transform backwards on
//                        conjugated inputs); summed in long double from the exact angles
inline void forward_tables(int sub, std::vector<cf>& tn, std::vector<cf>* lutc = nullptr) {
    tn.resize((size_t)sub * NPOLY * M_SUB);
    if (lutc) {
        lutc->resize((size_t)sub * NPOLY * 256);
        const long double tp = 6.283185307179586476925286766559005768L;
        for (int r = 0; r < sub; ++r)
            for (int ka = 0; ka < NPOLY; ++ka) {
                long double c[NPOLY], sn[NPOLY];
                for (int nu = 0; nu < NPOLY; ++nu) {
                    long long num = ((long long)ka * sub + r) * nu % ((long long)NPOLY * sub);
                    const long double th = tp * (long double)num / (long double)(NPOLY * sub);
                    c[nu] = cosl(th);
                    sn[nu] = -sinl(th);
                }
                for (int b = 0; b < 256; ++b) {
                    long double re = 0, im = 0;
                    for (int nu = 0; nu < NPOLY; ++nu) {
                        const long double sg = ((b >> nu) & 1) ? -1.0L : 1.0L;
                        re += sg * c[nu];
                        im += sg * sn[nu];
                    }
                    (*lutc)[((size_t)r * NPOLY + ka) * 256 + b] = mk((float)re, (float)(-im));
                }
            }
    }
    for (int r = 0; r < sub; ++r)
        for (int ka = 0; ka < NPOLY; ++ka) {
            const long long m = (long long)ka * sub + r;  // (kappa + r/sub) * sub
            for (int n = 0; n < M_SUB; ++n) tn[((size_t)r * NPOLY + ka) * M_SUB + n] = unit_fwd((long long)n * m, (long long)N_FFT * sub);
        }
}

// ---- C/A code ------------------------------------------------------------------------
// PRN -> G2 tap pair, c/search_offline.cpp:20-53 (the navstar column is unused there).
static const int kTaps[32][2] = {{2, 6}, {3, 7}, {4, 8},  {5, 9},  {1, 9}, {2, 10}, {1, 8}, {2, 9}, {3, 10}, {2, 3}, {3, 4},
                                 {5, 6}, {6, 7}, {7, 8},  {8, 9},  {9, 10}, {1, 4}, {2, 5}, {3, 6}, {4, 7},  {5, 8}, {6, 9},
                                 {1, 3}, {4, 6}, {5, 7},  {6, 8},  {7, 9}, {8, 10}, {1, 6}, {2, 7}, {3, 8},  {4, 9}};

// Gold-code generator as two 10-bit shift registers held in integers (bit i-1 = stage i).
class CaCode {
  public:
    CaCode(int t1, int t2) : g1_(0x3ff), g2_(0x3ff), m1_(1u << (t1 - 1)), m2_(1u << (t2 - 1)) {}
    int chip() const { return (int)(((g1_ >> 9) ^ ((g2_ & m1_) ? 1u : 0u) ^ ((g2_ & m2_) ? 1u : 0u)) & 1u); }
    void clock() {
        unsigned f1 = ((g1_ >> 2) ^ (g1_ >> 9)) & 1u;                                                     // stages 3,10
        unsigned f2 = ((g2_ >> 1) ^ (g2_ >> 2) ^ (g2_ >> 5) ^ (g2_ >> 7) ^ (g2_ >> 8) ^ (g2_ >> 9)) & 1u;  // 2,3,6,8,9,10
        g1_ = ((g1_ << 1) | f1) & 0x3ffu;
        g2_ = ((g2_ << 1) | f2) & 0x3ffu;
    }
    // same value as the reference's GetG1() (cacode.h:30-34): stage 10 is the MSB... of a 10-bit word
    unsigned g1_word() const {
        unsigned r = 0;
        for (int bit = 0; bit < 10; ++bit) r = 2 * r + ((g1_ >> (9 - bit)) & 1u);
        return r;
    }

  private:
    unsigned g1_, g2_, m1_, m2_;
};

// SearchCode(), c/search_offline.cpp:205-209
inline int search_code(int sv, int g1) {
    CaCode ca(kTaps[sv][0], kTaps[sv][1]);
    int chips = 0;
    while (ca.g1_word() != (unsigned)g1) {
        ca.clock();
        if (++chips > 2048) return -1;
    }
    return chips;
}

// Resampled code replica of SearchInit (c/search_offline.cpp:76,83-103).  The float/double
// promotions follow the reference's C expressions: ca_phase and chip are float, the literals
// 1.0 are double.
inline void code_replica(double fs, int sv, float* out) {
    const float ca_rate = (float)(1.023e6 / fs);
    CaCode ca(kTaps[sv][0], kTaps[sv][1]);
    float ca_phase = 0;
    for (int i = 0; i < N_FFT; ++i) {
        float chip = ca.chip() ? -1.0f : 1.0f;
        ca_phase += ca_rate;
        if (ca_phase >= 1.0) {
            ca_phase = (float)((double)ca_phase - 1.0);
            ca.clock();
            chip = (float)((double)chip * (1.0 - (double)ca_phase));
            chip = chip + ca_phase * (ca.chip() ? -1.0f : 1.0f);
        }
        out[i] = chip;
    }
}

// Quadrature LO as XOR masks in the capture's own bit packing (LSB first): bit i of
// cos_mask / sin_mask is lo_cos / lo_sin of int(lo_phase) at sample i
// (lo_sin = {1,1,0,0}, lo_cos = {0,1,1,0}; float accumulator, wrap at >= 4).
inline void lo_masks(double fc, double fs, int nbytes, uint8_t* cos_mask, uint8_t* sin_mask) {
    static const int lo_sin[4] = {1, 1, 0, 0}, lo_cos[4] = {0, 1, 1, 0};
    const float lo_rate = (float)(4 * fc / fs);
    float lo_phase = 0;
    std::memset(cos_mask, 0, nbytes);
    std::memset(sin_mask, 0, nbytes);
    for (int i = 0; i < nbytes * 8; ++i) {
        const int qd = (int)lo_phase;
        cos_mask[i >> 3] |= (uint8_t)(lo_cos[qd] << (i & 7));
        sin_mask[i >> 3] |= (uint8_t)(lo_sin[qd] << (i & 7));
        lo_phase += lo_rate;
        if (lo_phase >= 4) lo_phase -= 4;
    }
}

// The same masks bit-transposed for the forward kernel: byte n' of word B = n' / 8 holds, at bit nu,
// the mask bit of sample n' + 5000 nu (see acq_phases.hpp, fwd_stage_bits).
inline void transpose_masks(const uint8_t* mask /* >= 5000 bytes */, uint64_t* out /* [625] */) {
    for (int B = 0; B < 625; ++B) {
        uint64_t x = 0;
        for (int nu = 0; nu < 8; ++nu) x |= (uint64_t)mask[B + 625 * nu] << (8 * nu);
        uint64_t t;
        t = (x ^ (x >> 7)) & 0x00AA00AA00AA00AAull;  x = x ^ t ^ (t << 7);
        t = (x ^ (x >> 14)) & 0x0000CCCC0000CCCCull; x = x ^ t ^ (t << 14);
        t = (x ^ (x >> 28)) & 0x00000000F0F0F0F0ull; x = x ^ t ^ (t << 28);
        out[B] = x;
    }
}

// Search grid of Correlate(): Doppler half-range in bins (:176) and lags scanned (:190).
inline int doppler_half_range(double fs, double max_fo) { return (int)(max_fo * (double)N_FFT / fs); }
inline int num_lags(double fs) {
    int i = 0;
    while ((double)i < fs / 1000 && i < N_FFT) ++i;
    return i;
}

}  // namespace acq
