// acq_math.hpp -- butterfly math and index maps of the gfx950 acquisition kernels.
//
// The length-40000 transforms of the reference (fftwf_execute at c/search_offline.cpp:161,187)
// are computed here as 8 sub-transforms of length M = 5000 = 10 x 25 x 20 that live entirely in
// one workgroup's LDS (40 KB), plus one radix-8 step (decimation in time for the inverse, whose
// outputs are pruned; decimation in frequency for the forward, whose output rows are independent):
//
//   inverse (Correlate, :187)  y[n] = sum_{q<8} W_N^{-q n} F_q[n mod 5000],
//                              F_q  = IDFT_5000( prod[8 j + q] ),  only n < FS/1000 is formed
//   forward (Sample, :161)     X[8 k' + kappa] = DFT_5000( W_N^{n' kappa} sum_{nu<8} x[n' + 5000 nu] W_8^{nu kappa} )[k']
//
// Each length-5000 transform is three in-place LDS passes (decimation in frequency):
//   pass 1  radix-10 (Good-Thomas 2x5) on elements j' + 500 a,      twiddle W_5000^{j' alpha}
//   pass 2  radix-25 (5x5)            on elements j'' + 20 b,       twiddle W_500^{j'' beta}
//   pass 3  radix-20 (Good-Thomas 4x5) on elements j'',             no twiddle
// with LDS slot(alpha, j'', b) = 500 alpha + 25 j'' + b, and outputs r = 250 n'' + 10 beta + alpha.
//
// Complex values are 2-vectors of float so that on gfx950 every complex add is one
// v_pk_add_f32 and every complex multiply two packed instructions: the swap/negate forms
// (multiply by +-i, complex product) are written with VOP3P op_sel / neg modifiers in inline
// asm because hipcc otherwise spends a v_xor + v_mov per swap.  A wave64 VALU instruction
// holds the pipe 2 cycles (4 for v_pk_*), but one wave can only issue one every ~5 cycles
// (tools/ubench/valu_rate.hip), so at 2-4 waves per SIMD packed math is what fills the pipe.
//
// The functions are plain inline C++ outside device code so that tests/emul can run exactly
// this index math on the CPU (test infrastructure); the product only ever calls them from
// the HIP kernels.
#pragma once

#if defined(__HIPCC__)
#define ACQ_HD __host__ __device__ __forceinline__
#else
#define ACQ_HD inline
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define ACQ_PK_ASM 1
#else
#define ACQ_PK_ASM 0
#endif

namespace acq {

constexpr int N_FFT = 40000;  // FFT_LEN, c/gps_offline.h:15
constexpr int NPOLY = 8;      // polyphase components
constexpr int M_SUB = 5000;   // sub-transform length
constexpr int RA = 10, RB = 25, RC = 20;
constexpr int NBF1 = M_SUB / RA;  // 500 pass-1 butterflies
constexpr int NBF2 = M_SUB / RB;  // 200 pass-2 butterflies
constexpr int NBF3 = M_SUB / RC;  // 250 pass-3 butterflies (also the output column height)
constexpr int NW160 = N_FFT / NBF3;  // 160
constexpr int NT2 = RB * RC;         // 500 pass-2 twiddles
constexpr int MC_MAX = 40;           // accumulator columns per pass: 10000 lags; more lags take more passes
constexpr int WQ_STRIDE = NW160;     // wq[q][m] = W_160^{q m}, all 160 columns (40000 lags)
constexpr int FOLD_Q = NT2 + RA * NW160;  // folded-rotation tables per sub-transform: 500 pass-2 twiddles + 10 x 160 accumulate factors

typedef float cf __attribute__((ext_vector_type(2)));  // (re, im); one 64-bit VGPR pair on the device

ACQ_HD cf mk(float x, float y) { cf r; r.x = x; r.y = y; return r; }

// a * w
ACQ_HD cf cmul(cf a, cf w) {
#if ACQ_PK_ASM
    cf r;  // t = (ax wx, ax wy);  r = (t.x - ay wy, t.y + ay wx)
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]\n\t"
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=&v"(r) : "v"(a), "v"(w));
    return r;
#else
    return mk(a.x * w.x - a.y * w.y, a.x * w.y + a.y * w.x);
#endif
}
// a * conj(w)
ACQ_HD cf cmulc(cf a, cf w) {
#if ACQ_PK_ASM
    cf r;  // t = (ax wx, ay wx);  r = (t.x + ay wy, t.y - ax wy)
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]\n\t"
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]" : "=&v"(r) : "v"(a), "v"(w));
    return r;
#else
    return mk(a.x * w.x + a.y * w.y, a.y * w.x - a.x * w.y);
#endif
}
// same with a wave-uniform multiplier held in an SGPR pair (compile-time constants, scalar loads)
ACQ_HD cf cmul_u(cf a, cf w) {
#if ACQ_PK_ASM
    cf r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]\n\t"
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=&v"(r) : "v"(a), "s"(w));
    return r;
#else
    return cmul(a, w);
#endif
}
ACQ_HD cf cmulc_u(cf a, cf w) {
#if ACQ_PK_ASM
    cf r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]\n\t"
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]" : "=&v"(r) : "v"(a), "s"(w));
    return r;
#else
    return cmulc(a, w);
#endif
}
// acc + a * conj(w), w wave-uniform
ACQ_HD cf cmacc_u(cf acc, cf a, cf w) {
#if ACQ_PK_ASM
    cf r = acc;  // r += (ax wx, ay wx);  r += (ay wy, -ax wy)
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]\n\t"
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]" : "+v"(r) : "v"(a), "s"(w));
    return r;
#else
    return acc + cmulc(a, w);
#endif
}
// acc + a * conj(w), w per lane (VGPR pair)
ACQ_HD cf cmacc(cf acc, cf a, cf w) {
#if ACQ_PK_ASM
    cf r = acc;
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]\n\t"
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]" : "+v"(r) : "v"(a), "v"(w));
    return r;
#else
    return acc + cmulc(a, w);
#endif
}
// acc + a * w, w wave-uniform
ACQ_HD cf cmadd_u(cf acc, cf a, cf w) {
#if ACQ_PK_ASM
    cf r = acc;  // r += (ax wx, ax wy);  r += (-ay wy, ay wx)
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "+v"(r) : "v"(a), "s"(w));
    return r;
#else
    return acc + cmul(a, w);
#endif
}
// a + i b  and  a - i b
ACQ_HD cf add_i(cf a, cf b) {
#if ACQ_PK_ASM
    cf r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
#else
    return mk(a.x - b.y, a.y + b.x);
#endif
}
ACQ_HD cf sub_i(cf a, cf b) {
#if ACQ_PK_ASM
    cf r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
#else
    return mk(a.x + b.y, a.y - b.x);
#endif
}

// DIR = -1: forward transform (exp(-i..)), DIR = +1: backward.  Tables hold the forward value
// w = exp(-i theta); tw<DIR>(a, w) multiplies by w or conj(w).
template <int DIR> ACQ_HD cf tw(cf a, cf w) { return DIR < 0 ? cmul(a, w) : cmulc(a, w); }
template <int DIR> ACQ_HD cf tw_u(cf a, cf w) { return DIR < 0 ? cmul_u(a, w) : cmulc_u(a, w); }
// a + DIR*i*b, a - DIR*i*b
template <int DIR> ACQ_HD cf add_di(cf a, cf b) { return DIR > 0 ? add_i(a, b) : sub_i(a, b); }
template <int DIR> ACQ_HD cf sub_di(cf a, cf b) { return DIR > 0 ? sub_i(a, b) : add_i(a, b); }

template <int DIR> ACQ_HD void dft2(cf& a, cf& b) {
    cf t = a - b;
    a = a + b;
    b = t;
}

template <int DIR> ACQ_HD void dft4(cf& a, cf& b, cf& c, cf& d) {
    cf apc = a + c, amc = a - c, bpd = b + d, bmd = b - d;
    a = apc + bpd;
    b = add_di<DIR>(amc, bmd);
    c = apc - bpd;
    d = sub_di<DIR>(amc, bmd);
}

// a + i s b  and  a - i s b  (s real, wave-uniform: its value in both halves of an SGPR pair): one packed FMA
ACQ_HD cf fma_i(cf a, cf b, cf s) {
#if ACQ_PK_ASM
    cf r;  // (a.x - s b.y, a.y + s b.x)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,0,1] neg_lo:[1,0,0]" : "=v"(r) : "v"(b), "s"(s), "v"(a));
    return r;
#else
    return mk(a.x - s.x * b.y, a.y + s.x * b.x);
#endif
}
ACQ_HD cf fms_i(cf a, cf b, cf s) {
#if ACQ_PK_ASM
    cf r;  // (a.x + s b.y, a.y - s b.x)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,0,1] neg_hi:[1,0,0]" : "=v"(r) : "v"(b), "s"(s), "v"(a));
    return r;
#else
    return mk(a.x + s.x * b.y, a.y - s.x * b.x);
#endif
}
template <int DIR> ACQ_HD cf fma_di(cf a, cf b, cf s) { return DIR > 0 ? fma_i(a, b, s) : fms_i(a, b, s); }
template <int DIR> ACQ_HD cf fms_di(cf a, cf b, cf s) { return DIR > 0 ? fms_i(a, b, s) : fma_i(a, b, s); }

// Five-point butterfly.  With c1 = cos(2 pi/5), c2 = cos(4 pi/5), s1 = sin(2 pi/5), s2 = sin(4 pi/5):
//   X0 = x0 + t1 + t2,  X1,4 = m1 +- i DIR sg1,  X2,3 = m2 +- i DIR sg2,
//   m1 = x0 + c1 t1 + c2 t2,  m2 = x0 + c2 t1 + c1 t2,  sg1 = s1 t3 + s2 t4,  sg2 = s2 t3 - s1 t4.
// Formed in 15 packed instructions, 9 of them FMAs (round 4; 18 before): the cosine part through u = t1 + t2, w = x0 + c2 u,
// m1 = w + (c1 - c2) t1, m2 = w + (c1 - c2) t2 (5 instead of 6), the sine part scaled by 1/s1 -- sg1/s1 = t3 + (s2/s1) t4,
// sg2/s1 = (s2/s1) t3 - t4 (2 instead of 4) -- with s1 restored inside the +-i add that forms the outputs (a packed FMA with
// op_sel / neg modifiers instead of a packed add).
template <int DIR> ACQ_HD void dft5(cf& x0, cf& x1, cf& x2, cf& x3, cf& x4) {
    constexpr float C2 = -0.8090169943749473f, S1 = 0.9510565162951535f;
    cf t1 = x1 + x4, t2 = x2 + x3, t3 = x1 - x4, t4 = x2 - x3;
    constexpr float CD = 1.118033988749895f;   // c1 - c2 = sqrt(5)/2
    constexpr float SR = 0.6180339887498949f;  // s2 / s1
    const cf s1v = mk(S1, S1);
    cf u = t1 + t2;
    cf w = x0 + C2 * u;
    x0 = x0 + u;
    cf m1 = w + CD * t1;
    cf m2 = w + CD * t2;
    cf g1 = t3 + SR * t4;
    cf g2 = SR * t3 - t4;
    x1 = fma_di<DIR>(m1, g1, s1v);
    x4 = fms_di<DIR>(m1, g1, s1v);
    x2 = fma_di<DIR>(m2, g2, s1v);
    x3 = fms_di<DIR>(m2, g2, s1v);
}

// forward value of W_8^m = exp(-2 pi i m / 8), m taken mod 8
ACQ_HD cf w8(int m) {
    constexpr float R = 0.7071067811865476f;
    switch (m & 7) {
        case 0: return mk(1.f, 0.f);
        case 1: return mk(R, -R);
        case 2: return mk(0.f, -1.f);
        case 3: return mk(-R, -R);
        case 4: return mk(-1.f, 0.f);
        case 5: return mk(-R, R);
        case 6: return mk(0.f, 1.f);
        default: return mk(R, R);
    }
}
// Output kappa of an 8-point forward DFT: sum_nu x[nu] W_8^{nu kappa}, with the three wave-uniform
// rotations c4 = W_8^{4 kappa} (= +-1), c2 = W_8^{2 kappa}, c1 = W_8^{kappa} supplied by the caller.
ACQ_HD cf dft8_one(const cf* x, cf c4, cf c2, cf c1) {
    cf u0 = x[0] + c4.x * x[4], u1 = x[1] + c4.x * x[5], u2 = x[2] + c4.x * x[6], u3 = x[3] + c4.x * x[7];
    cf v0 = cmadd_u(u0, u2, c2), v1 = cmadd_u(u1, u3, c2);
    return cmadd_u(v0, v1, c1);
}

// 8-point DFT, natural order in and out.
template <int DIR> ACQ_HD void dft8(cf* x) {
    constexpr float R = 0.7071067811865476f;
    cf e0 = x[0], e1 = x[2], e2 = x[4], e3 = x[6];
    cf o0 = x[1], o1 = x[3], o2 = x[5], o3 = x[7];
    dft4<DIR>(e0, e1, e2, e3);
    dft4<DIR>(o0, o1, o2, o3);
    // o_k *= W_8^{k} (forward exp(-i pi k/4)); the k = 2 factor (-+i) is folded into the adds
    o1 = tw_u<DIR>(o1, mk(R, -R));
    o3 = tw_u<DIR>(o3, mk(-R, -R));
    x[0] = e0 + o0; x[4] = e0 - o0;
    x[1] = e1 + o1; x[5] = e1 - o1;
    x[2] = add_di<DIR>(e2, o2); x[6] = sub_di<DIR>(e2, o2);
    x[3] = e3 + o3; x[7] = e3 - o3;
}

// 10 = 2 x 5 prime-factor butterfly: n = (5 n1 + 2 n2) mod 10, k = (5 k1 + 6 k2) mod 10.
template <int DIR> ACQ_HD void radix10(const cf* x, cf* y) {
    cf u[2][5];
#pragma unroll
    for (int n2 = 0; n2 < 5; ++n2) {
        cf a = x[(2 * n2) % 10], b = x[(5 + 2 * n2) % 10];
        dft2<DIR>(a, b);
        u[0][n2] = a;
        u[1][n2] = b;
    }
#pragma unroll
    for (int k1 = 0; k1 < 2; ++k1) {
        dft5<DIR>(u[k1][0], u[k1][1], u[k1][2], u[k1][3], u[k1][4]);
#pragma unroll
        for (int k2 = 0; k2 < 5; ++k2) y[(5 * k1 + 6 * k2) % 10] = u[k1][k2];
    }
}

// 20 = 4 x 5 prime-factor butterfly: n = (5 n1 + 4 n2) mod 20, k = (5 k1 + 16 k2) mod 20.
template <int DIR> ACQ_HD void radix20(const cf* x, cf* y) {
    cf u[4][5];
#pragma unroll
    for (int n2 = 0; n2 < 5; ++n2) {
        cf a = x[(4 * n2) % 20], b = x[(5 + 4 * n2) % 20], c = x[(10 + 4 * n2) % 20], d = x[(15 + 4 * n2) % 20];
        dft4<DIR>(a, b, c, d);
        u[0][n2] = a; u[1][n2] = b; u[2][n2] = c; u[3][n2] = d;
    }
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) {
        dft5<DIR>(u[k1][0], u[k1][1], u[k1][2], u[k1][3], u[k1][4]);
#pragma unroll
        for (int k2 = 0; k2 < 5; ++k2) y[(5 * k1 + 16 * k2) % 20] = u[k1][k2];
    }
}

// cos / sin of 2 pi m / 25 in double, for the compile-time constants of dft5_tw
constexpr double w25cos(int m) {
    return m == 1 ? 0.96858316112863108 : m == 2 ? 0.87630668004386358 : m == 3 ? 0.72896862742141155 : m == 4 ? 0.53582679497899655
         : m == 6 ? 0.062790519529313527 : m == 8 ? -0.42577929156507272 : m == 9 ? -0.63742398974868975 : m == 12 ? -0.99211470131447776
                                                                                                                    : -0.63742398974868952;  // 16
}
constexpr double w25sin(int m) {
    return m == 1 ? 0.24868988716485479 : m == 2 ? 0.48175367410171532 : m == 3 ? 0.68454710592868862 : m == 4 ? 0.84432792550201508
         : m == 6 ? 0.99802672842827156 : m == 8 ? 0.90482705246601947 : m == 9 ? 0.77051324277578925 : m == 12 ? 0.12533323356430454
                                                                                                                  : -0.77051324277578936;  // 16
}
// Second-stage butterfly of the radix-25, inner twiddles included: the five-point transform of (x0, x1 w^1, x2 w^2, x3 w^3, x4 w^4)
// with w^j = W_25^{K1 j} (conjugated for DIR > 0), in 19 packed instructions instead of 8 + 15.  A twiddle w = c (1 -+ i t), t = tan:
// y_j = x_j -+ i t_j x_j is ONE packed FMA (fma_i / fms_i with both vector operands x_j) and leaves the real scale c_j pending; all
// constants being known at compile time, the pending scales cost nothing in the butterfly -- every add of two differently scaled
// terms was going to be an FMA or becomes one with the ratio of the scales as its constant:
//   t1' = y1 + (c4/c1) y4, t3' = y1 - (c4/c1) y4 (= t1/c1, t3/c1);  t2' = y2 + (c3/c2) y3, t4' = y2 - (c3/c2) y3 (= t2/c2, t4/c2)
//   u' = t1' + (c2/c1) t2';  X0 = x0 + c1 u';  w = x0 + (C2 c1) u';  m1 = w + (CD c1) t1';  m2 = w + (CD c2) t2'
//   g1' = t3' + (SR c2/c1) t4';  g2' = (SR c1/c2) t3' - t4';  X1,4 = m1 +- i (S1 c1) g1';  X2,3 = m2 +- i (S1 c2) g2'
// (C2, CD, SR, S1 as in dft5).  |t| <= 15.9 (m = 6): products stay within a few ulp of the plain form's (tests/emul, GPU parity).
template <int DIR, int K1> ACQ_HD void dft5_tw(cf& x0, cf& x1, cf& x2, cf& x3, cf& x4) {
    constexpr double c1 = w25cos(K1), c2 = w25cos(2 * K1), c3 = w25cos(3 * K1), c4 = w25cos(4 * K1);
    constexpr double C2 = -0.8090169943749473, CD = 1.118033988749895, SR = 0.6180339887498949, S1 = 0.9510565162951535;
    constexpr float T1 = (float)(w25sin(K1) / c1), T2 = (float)(w25sin(2 * K1) / c2), T3 = (float)(w25sin(3 * K1) / c3), T4 = (float)(w25sin(4 * K1) / c4);
    constexpr float R14 = (float)(c4 / c1), R23 = (float)(c3 / c2), R21 = (float)(c2 / c1), K0 = (float)c1, KW = (float)(C2 * c1);
    constexpr float KM1 = (float)(CD * c1), KM2 = (float)(CD * c2), KG1 = (float)(SR * c2 / c1), KG2 = (float)(SR * c1 / c2);
    constexpr float KS1 = (float)(S1 * c1), KS2 = (float)(S1 * c2);
    // forward: x w = c (x - i t x); backward: x conj(w) = c (x + i t x)
    cf y1 = DIR < 0 ? fms_i(x1, x1, mk(T1, T1)) : fma_i(x1, x1, mk(T1, T1));
    cf y2 = DIR < 0 ? fms_i(x2, x2, mk(T2, T2)) : fma_i(x2, x2, mk(T2, T2));
    cf y3 = DIR < 0 ? fms_i(x3, x3, mk(T3, T3)) : fma_i(x3, x3, mk(T3, T3));
    cf y4 = DIR < 0 ? fms_i(x4, x4, mk(T4, T4)) : fma_i(x4, x4, mk(T4, T4));
    cf t1 = y1 + R14 * y4, t3 = y1 - R14 * y4, t2 = y2 + R23 * y3, t4 = y2 - R23 * y3;
    cf u = t1 + R21 * t2;
    cf w = x0 + KW * u;
    x0 = x0 + K0 * u;
    cf m1 = w + KM1 * t1;
    cf m2 = w + KM2 * t2;
    cf g1 = t3 + KG1 * t4;
    cf g2 = KG2 * t3 - t4;
    x1 = fma_di<DIR>(m1, g1, mk(KS1, KS1));
    x4 = fms_di<DIR>(m1, g1, mk(KS1, KS1));
    x2 = fma_di<DIR>(m2, g2, mk(KS2, KS2));
    x3 = fms_di<DIR>(m2, g2, mk(KS2, KS2));
}

// 25 = 5 x 5 Cooley-Tukey butterfly: input n = 5 n1 + n2, output k = k1 + 5 k2.
template <int DIR> ACQ_HD void radix25(const cf* x, cf* y) {
    cf v[5][5];  // v[k1][n2]
#pragma unroll
    for (int n2 = 0; n2 < 5; ++n2) {
        cf a = x[n2], b = x[5 + n2], c = x[10 + n2], d = x[15 + n2], e = x[20 + n2];
        dft5<DIR>(a, b, c, d, e);
        v[0][n2] = a; v[1][n2] = b; v[2][n2] = c; v[3][n2] = d; v[4][n2] = e;
    }
    dft5<DIR>(v[0][0], v[0][1], v[0][2], v[0][3], v[0][4]);
    dft5_tw<DIR, 1>(v[1][0], v[1][1], v[1][2], v[1][3], v[1][4]);
    dft5_tw<DIR, 2>(v[2][0], v[2][1], v[2][2], v[2][3], v[2][4]);
    dft5_tw<DIR, 3>(v[3][0], v[3][1], v[3][2], v[3][3], v[3][4]);
    dft5_tw<DIR, 4>(v[4][0], v[4][1], v[4][2], v[4][3], v[4][4]);
#pragma unroll
    for (int k1 = 0; k1 < 5; ++k1)
#pragma unroll
        for (int k2 = 0; k2 < 5; ++k2) y[k1 + 5 * k2] = v[k1][k2];
}

// ---------------------------------------------------------------------------------------
// LDS passes of one length-5000 transform.  Twiddle tables (acq_tables.hpp):
//   t1[alpha*500 + j']  = W_5000^{j' alpha}      pass-1 outputs (q-independent; the correlator
//                                                keeps its 18 values per thread in registers)
//   t2[beta*20 + j'']   = W_500^{j'' beta}       pass-2 outputs (q-independent, 4 KB;
//                                                the correlator keeps the table in LDS)
//   bq[q*250 + t3]      = W_40000^{q rho(t3)}    rotation of sub-transform q at pass-3 thread t3
//   wq[q*40 + m]        = W_160^{q m}            the wave-uniform rest of W_40000^{q n}, n = 250 m + rho

// two consecutive complex values with one 16-byte access (p 16-byte aligned / only 8-byte aligned)
typedef float cf2 __attribute__((ext_vector_type(4)));
typedef cf2 cf2_a8 __attribute__((aligned(8)));
ACQ_HD void ld2(const cf* p, cf& a, cf& b) {
    const cf2 v = *reinterpret_cast<const cf2*>(p);
    a = v.xy;
    b = v.zw;
}
ACQ_HD void ld2u(const cf* p, cf& a, cf& b) {
    const cf2 v = *reinterpret_cast<const cf2_a8*>(p);
    a = v.xy;
    b = v.zw;
}

// Where element (alpha, j'', b) of the in-place transform lives in the workgroup's LDS buffer (complex units):
//   LayA (k_fwd)           500 alpha + 25 j'' + b : a pass-2 butterfly's 25 elements are contiguous (paired 8-byte accesses),
//                          pass-1 stores and pass-3 reads are strided and pay 2-way bank conflicts
//   LayB (k_corr)          564 alpha + 22 b + j'' : the two butterflies a pass-1 thread owns (j' = 2 tid, 2 tid + 1) are
//                          neighbours, so their outputs leave as ONE 16-byte store per alpha, and a pass-3 thread's 20
//                          elements are contiguous (ten 16-byte reads); 22 and 564 keep every access conflict-free
//                          (176-byte lane stride for the 16-byte reads, 564 - 20 = 17 * 32 across the alpha boundary of pass 2)
//   LayC (k_corr, round 3)  LayB's slots with the WORK re-dealt over the lanes so that every LDS instruction of the three passes is
//                          bank-conflict free (the hardware services a wave's access in fixed lane groups -- 8 contiguous lanes for
//                          16-byte stores, 16 for 8-byte accesses, the interleaved 16-lane sets of ds_read_b128 -- and only lanes of
//                          one group can collide, MI355X_MICROARCH.md section LDS):
//                            pass 1  thread t owns the butterfly pair (b, j), j' = 20 b + 2 j: t < 200 -> (t / 8, t % 8); the pairs
//                                    j = 8, 9 of four rows b with distinct (3 b) mod 8 share a group of 8 (pass1_jp)
//                            pass 2  lane e < 160 -> (alpha, j'') = (e / 16, e % 16); the j'' = 16..19 of four alpha share a group
//                                    (their twiddle reads then hit ONE address per j'': a broadcast) (pass2_owner)
//                            pass 3  thread -> rho through a 250-entry table (kRhoC) that gives every 16-lane read group 16
//                                    distinct bank classes (10 alpha + 11 beta) mod 16
//                          Same elements, same arithmetic per element: passes 1 and 2 are bit-identical to LayB, pass 3 only
//                          changes which thread accumulates which lag.  LDS-array cycles per sub-transform 1930 -> 1462 in the
//                          instruction-level model that reproduces the measured conflict count of LayB (488) exactly.
struct LayA { static constexpr int SA = NBF1, SJ = RB, SB = 1, SIZE = M_SUB; static constexpr bool REMAP = false; };
struct LayB { static constexpr int SA = 564, SJ = 1, SB = 22, SIZE = 10 * 564; static constexpr bool REMAP = false; };
struct LayC { static constexpr int SA = 564, SJ = 1, SB = 22, SIZE = 10 * 564; static constexpr bool REMAP = true; };

// LayC: thread t3 (< 250) of pass 3 -> rho = 10 beta + alpha of the radix-20 butterfly it owns (generated by tools/lds_maps.py)
static const unsigned char kRhoC[NBF3] = {
      0,  20,  40,  60, 160, 180, 200, 220, 170, 190, 210, 240,  10,  30,  50,  80,   1, 230, 121, 141, 140,  70, 100, 120, 150,
     90, 110, 130,  11, 111, 131, 151,  21,  41,  61,  81, 181, 201, 221, 241, 191, 211, 231, 122,  31,  51,  71, 101,  22, 112,
      3, 162, 161,  91, 142,   2, 171, 132, 152,  12,  32, 153,  13, 172,  42,  62,  82, 102, 202, 222, 242, 123, 212, 232, 113,
      4,  52,  72,  92, 143,  43, 154,  24, 183, 182, 133, 163,  23, 192,  14, 173,  33,  53, 174,  34, 193,  63,  83, 103, 144,
    223, 243, 124,   5, 233, 114, 155,  25,  73,  93, 134, 164,  64, 175,  45, 204, 203,  15, 184,  44, 213,  35, 194,  54,  74,
    195,  55, 214,  84, 104, 145, 165, 244, 125,   6,  26, 115, 156, 176,  46,  94, 135,  16, 185,  85, 196,  66, 225, 224,  36,
    205,  65, 234,  56, 215,  75,  95, 216,  76, 235, 105, 146, 166, 186, 126,   7,  27,  47, 157, 177, 197,  67, 136,  17,  37,
    206, 106, 217,  87, 246, 245,  57, 226,  86, 116,  77, 236,  96, 137, 237,  97, 117, 147, 167, 187, 207,   8,  28,  48,  68,
    178, 198, 218,  88,  18,  38,  58, 227, 148, 238, 108, 128, 127,  78, 247, 107, 158,  98, 118, 138,  19, 119, 139, 159, 168,
    188, 208, 228,  49,  69,  89, 189, 199, 219, 239, 109,  39,  59,  79, 248, 169, 209, 229, 249,   9,  99, 129, 149, 179,  29};

// first (even) butterfly j' of the pair pass-1 thread t owns
template <class L> ACQ_HD int pass1_jp(int t) {
    if (!L::REMAP) return 2 * t;
    if (t < 200) return RC * (t >> 3) + 2 * (t & 7);
    const int r = t - 200, g = r >> 3, i = r & 7;  // the pairs j = 8, 9: rows {0,2,4,6}+8g, then {1,3,5,7}+8(g-3), then row 24
    const int b = g < 3 ? 8 * g + 2 * (i >> 1) : (g < 6 ? 8 * (g - 3) + 2 * (i >> 1) + 1 : 24);
    return RC * b + 16 + 2 * (i & 1);
}
// (alpha, j'') of the radix-25 butterfly pass-2 lane e (< 200) owns
template <class L> ACQ_HD void pass2_owner(int e, int& al, int& jpp) {
    if (!L::REMAP) {
        al = e / RC;
        jpp = e - al * RC;
    } else if (e < 160) {
        al = e >> 4;
        jpp = e & 15;
    } else {
        al = (e - 160) >> 2;
        jpp = 16 + (e & 3);
    }
}

// pass 1 for butterfly jp (0..499): x[a] = element jp + 500 a; w[al-1] = t1[al*500 + jp].
template <int DIR, class L = LayA> ACQ_HD void pass1_store(const cf* x, int jp, const cf* w, cf* lds) {
    cf y[RA];
    radix10<DIR>(x, y);
    const int b = jp / RC, jpp = jp - b * RC;
    cf* dst = lds + L::SJ * jpp + L::SB * b;
    dst[0] = y[0];
#pragma unroll
    for (int al = 1; al < RA; ++al) dst[L::SA * al] = tw<DIR>(y[al], w[al - 1]);
}
// forward value of W_5000^al, al = 1..9: the ratio of the pass-1 twiddles of two neighbouring butterflies
template <int AL> ACQ_HD cf w5000() {
    static_assert(AL >= 1 && AL <= 9, "w5000");
    return AL == 1   ? mk(0.99999921043206784f, -0.0012566367307361701f)
           : AL == 2 ? mk(0.99999684172951656f, -0.0025132714770690740f)
           : AL == 3 ? mk(0.99999289389607086f, -0.0037699022546219146f)
           : AL == 4 ? mk(0.99998736693796492f, -0.0050265270790207270f)
           : AL == 5 ? mk(0.99998026086392511f, -0.0062831439655589511f)
           : AL == 6 ? mk(0.99997157568517318f, -0.0075397509301827640f)
           : AL == 7 ? mk(0.99996131141542431f, -0.0087963459885950153f)
           : AL == 8 ? mk(0.99994946807088674f, -0.010052927156601628f)
                     : mk(0.99993604567026163f, -0.011309492350427326f);
}
// LayB / LayC: butterflies jp and jp + 1 (jp even) together, one 16-byte store per alpha.  w0[al-1] = W_5000^{jp al}.
// W1H = false: w1[al-1] = W_5000^{(jp + 1) al} comes from registers too (36 twiddle registers per thread);
// W1H = true: it is formed as w0 * W_5000^al (wave-uniform constant): 9 more complex multiplies per sub-transform,
// 18 registers fewer -- what lets the 33-column instance run three workgroups per CU.
template <int DIR, bool W1H, int AL> ACQ_HD void pass1_pair_one(const cf* y0, const cf* y1, const cf* w0, const cf* w1, cf* dst) {
    cf a0, a1;
    if (AL == 0) {
        a0 = y0[0];
        a1 = y1[0];
    } else {
        a0 = tw<DIR>(y0[AL], w0[AL - 1]);
        a1 = W1H ? tw<DIR>(tw_u<DIR>(y1[AL], w5000<(AL ? AL : 1)>()), w0[AL - 1]) : tw<DIR>(y1[AL], w1[AL - 1]);
    }
    cf2 v;
    v.xy = a0;
    v.zw = a1;
    *reinterpret_cast<cf2*>(dst + LayB::SA * AL) = v;
}
template <int DIR, bool W1H = false> ACQ_HD void pass1_store_pair(const cf* x0, const cf* x1, int jp, const cf* w0, const cf* w1, cf* lds) {
    cf y0[RA], y1[RA];
    radix10<DIR>(x0, y0);
    radix10<DIR>(x1, y1);
    const int b = jp / RC, jpp = jp - b * RC;  // jp, hence jpp, even: the pair never straddles a row of 20
    cf* dst = lds + LayB::SB * b + jpp;
    pass1_pair_one<DIR, W1H, 0>(y0, y1, w0, w1, dst);
    pass1_pair_one<DIR, W1H, 1>(y0, y1, w0, w1, dst);
    pass1_pair_one<DIR, W1H, 2>(y0, y1, w0, w1, dst);
    pass1_pair_one<DIR, W1H, 3>(y0, y1, w0, w1, dst);
    pass1_pair_one<DIR, W1H, 4>(y0, y1, w0, w1, dst);
    pass1_pair_one<DIR, W1H, 5>(y0, y1, w0, w1, dst);
    pass1_pair_one<DIR, W1H, 6>(y0, y1, w0, w1, dst);
    pass1_pair_one<DIR, W1H, 7>(y0, y1, w0, w1, dst);
    pass1_pair_one<DIR, W1H, 8>(y0, y1, w0, w1, dst);
    pass1_pair_one<DIR, W1H, 9>(y0, y1, w0, w1, dst);
}

// pass 2 for butterfly e (0..199), in place; t2 may live in LDS (its own allocation, so the
// compiler knows the in-place stores do not alias it) or in global memory.
template <int DIR, class L = LayA> ACQ_HD void pass2_inplace(int e, const cf* __restrict__ t2, cf* lds) {
    int al, jpp;
    pass2_owner<L>(e, al, jpp);
    cf* p = lds + L::SA * al + L::SJ * jpp;
    cf x[RB], y[RB];
#pragma unroll
    for (int b = 0; b < RB; ++b) x[b] = p[L::SB * b];
    radix25<DIR>(x, y);
    p[0] = y[0];
#pragma unroll
    for (int be = 1; be < RB; ++be) p[L::SB * be] = tw<DIR>(y[be], t2[be * RC + jpp]);
}

// pass 3 for the butterfly rho = 10 beta + alpha (0..249): y[n''] = F[250 n'' + rho].
template <int DIR, class L = LayA> ACQ_HD void pass3_load(int rho, const cf* lds, cf* y) {
    const int be = rho / RA, al = rho - be * RA;
    const cf* p = lds + L::SA * al + L::SB * be;
    cf x[RC];
    if (L::SJ == 1) {  // contiguous: 16-byte reads (SA and SB keep p 16-byte aligned)
#pragma unroll
        for (int jpp = 0; jpp < RC; jpp += 2) ld2(p + jpp, x[jpp], x[jpp + 1]);
    } else {
#pragma unroll
        for (int jpp = 0; jpp < RC; ++jpp) x[jpp] = p[L::SJ * jpp];
    }
    radix20<DIR>(x, y);
}
// which butterfly pass-3 thread t3 (< 250) owns
template <class L = LayA> ACQ_HD int pass3_rho(int t3) {
    if (L::REMAP) return kRhoC[t3];  // (host / emulation; the kernels read the same table from device memory)
    const int al = t3 / RB, be = t3 - al * RB;
    return RA * be + al;
}

// Doppler shift of the code spectrum by whole bins (c/search_offline.cpp:182):
// C[(8 j + q - dop) mod N] = code_pp[q'][(j + c) mod 5000] with
// q' = (q - dop) mod 8 (non-negative), c = floor((q - dop) / 8).
ACQ_HD void shift_split(int q, int dop, int& qp, int& c) {
    const int e = q - dop;
    qp = e & 7;
    c = (e - qp) >> 3;  // exact: e - qp is a multiple of 8
}

}  // namespace acq
