// acq_corr8.hpp -- the 8-wave correlator (round 3): per-thread phase bodies of k_corr8, the variant of Correlate()'s cell
// kernel (c/search_offline.cpp:181-196) built for OCCUPANCY instead of few passes.
//
// k_corr (acq_phases.hpp) runs a 5000-point sub-transform as 10 x 25 x 20 on 250 threads: 20-25 complex values per thread and
// pass, 44 accumulator + 36 twiddle registers held across the radix-25 -> 154 VGPRs, three 4-wave workgroups per CU.  The
// kernel responds to occupancy and to little else.  Here the same transform is 5 x 10 x 10 x 10 on 500
// threads (8 waves): ten values per thread and pass, 11 accumulators per thread, pass-2/3 twiddles in LDS --
// fewer live registers, hence more resident waves, at the price of one more LDS round trip and barrier per sub-transform:
//
//   j = 1000 a + 100 b + 10 c + d           input index of the sub-transform (a < 5; b, c, d < 10)
//   r = alpha + 5 beta + 50 gamma + 500 delta   output index (decimation in frequency: pass k turns digit k into its output digit)
//   pass 1  radix 5  over a   thread u: the two neighbouring columns j' = 2u, 2u+1 (16-byte loads of both spectra, product
//                             conj(D) C formed in registers); twiddle W_5000^{j' alpha} from registers
//   pass 2  radix 10 over b   thread (alpha, 10 c + d);  twiddle W_1000^{(10 c + d) beta} from an LDS table
//   pass 3  radix 10 over c   thread (alpha, beta, d);   twiddle W_100^{d gamma} from an LDS table
//   pass 4  radix 10 over d   thread rho = alpha + 5 beta + 50 gamma = tid: outputs r = rho + 500 delta, rotated by the
//                             polyphase factor W_N^{-q n} = conj(bq8[q][rho]) conj(W_80^{q m}), n = 500 m + rho, and
//                             accumulated over the 8 polyphase components in registers (only the FS/1000 lags are formed)
// LDS slot of element (alpha, beta, gamma, delta) -- digits in whatever stage they are --: 1024 alpha + 102 beta + 10 gamma + delta.
//
// Plain inline C++ outside device code like acq_phases.hpp, so that tests/emul runs exactly this index math on the CPU.
#pragma once
#include "acq_phases.hpp"

namespace acq {

constexpr int WG8 = 512;        // workgroup size of k_corr8
constexpr int NT8 = 500;        // threads that own work
constexpr int R8A = 5, R8 = 10;
struct Lay8 {
    static constexpr int SA = 1024, SB = 102, SC = 10, SIZE = R8A * SA;  // complex elements (40 KB)
};
constexpr int NT8_T2 = 9 * 100;  // pass-2 twiddles W_1000^{j'' beta}, beta = 1..9, j'' < 100: [beta - 1][j'']
constexpr int NT8_T3 = 9 * 10;   // pass-3 twiddles W_100^{d gamma}, gamma = 1..9, d < 10:      [gamma - 1][d]
constexpr int WQ8_STRIDE = N_FFT / 500;  // 80: wq8[q][m] = W_80^{q m}
constexpr int MC8_MAX = 20;              // accumulator columns: 10000 lags

ACQ_HD int slot8(int al, int be, int ga, int de) { return Lay8::SA * al + Lay8::SB * be + Lay8::SC * ga + de; }

// forward value of W_5000^al, al = 1..4: ratio of the pass-1 twiddles of two neighbouring columns
template <int AL> ACQ_HD cf w5000_8() { return w5000<AL>(); }

// pass-1 twiddles of thread u (columns j' = 2u, 2u + 1): w[0][al-1] = W_5000^{2u al}, w[1][al-1] = W_5000^{(2u+1) al}
// t1_8: [al][j'] = W_5000^{j' al}, al < 5, j' < 1000
// W1H: only column 2u's are held; column 2u+1's are formed as w * W_5000^al (wave-uniform constant) in pass 1
template <bool W1H = false>
ACQ_HD void load_tw8(int tid, const cf* __restrict__ t1_8, cf (&w)[2][R8A - 1]) {
    if (tid >= NT8) return;
#pragma unroll
    for (int al = 1; al < R8A; ++al) {
        if (W1H) w[0][al - 1] = t1_8[al * 1000 + 2 * tid];
        else ld2(t1_8 + al * 1000 + 2 * tid, w[0][al - 1], w[1][al - 1]);
    }
}
template <int DIR, bool W1H, int AL> ACQ_HD cf tw8_second(cf y, const cf (&w)[2][R8A - 1]) {
    return W1H ? tw<DIR>(tw_u<DIR>(y, w5000<(AL ? AL : 1)>()), w[0][AL - 1]) : tw<DIR>(y, w[1][AL - 1]);
}

// pass 1: loads, product conj(D) C (:181-185; D is stored conjugated), radix 5, twiddle, 16-byte stores
template <bool W1H = false>
ACQ_HD void corr8_phase1(int tid, int q, int dop, const cf* __restrict__ dpp, const cf* __restrict__ cpp, int crow, int halo,
                         const cf (&w)[2][R8A - 1], cf* lds) {
    if (tid >= NT8) return;
    const int jp = 2 * tid;
    int qp, c;
    shift_split(q, dop, qp, c);
    cf x0[R8A], x1[R8A];
#if defined(__HIP_DEVICE_COMPILE__)
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void*)dpp, 0, NPOLY * M_SUB * (int)sizeof(cf), 0x00020000);
    const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc((void*)cpp, 0, NPOLY * crow * (int)sizeof(cf), 0x00020000);
    const int sd = q * M_SUB * (int)sizeof(cf), sc = (qp * crow + halo + c) * (int)sizeof(cf);
    const int lane = jp * (int)sizeof(cf);
    cf2 d[R8A], cc[R8A];
#pragma unroll
    for (int a = 0; a < R8A; ++a) {
        constexpr int ROW = 1000 * (int)sizeof(cf);
        d[a] = __builtin_bit_cast(cf2, __builtin_amdgcn_raw_buffer_load_b128(rd, lane, sd + ROW * a, 0));
        cc[a] = __builtin_bit_cast(cf2, __builtin_amdgcn_raw_buffer_load_b128(rc, lane, sc + ROW * a, 0));
    }
#pragma unroll
    for (int a = 0; a < R8A; ++a) {
        x0[a] = cmul(d[a].xy, cc[a].xy);
        x1[a] = cmul(d[a].zw, cc[a].zw);
    }
#else
    const cf* drow = dpp + q * M_SUB + jp;
    const cf* crw = cpp + (long)qp * crow + halo + c + jp;
    for (int a = 0; a < R8A; ++a) {
        x0[a] = cmul(drow[1000 * a], crw[1000 * a]);
        x1[a] = cmul(drow[1000 * a + 1], crw[1000 * a + 1]);
    }
#endif
    dft5<+1>(x0[0], x0[1], x0[2], x0[3], x0[4]);
    dft5<+1>(x1[0], x1[1], x1[2], x1[3], x1[4]);
    const int b = jp / 100, r = jp - 100 * b, cdig = r / 10, ddig = r - 10 * cdig;  // ddig even: the pair shares (b, c)
    cf* dst = lds + slot8(0, b, cdig, ddig);
    cf2 v;
    v.xy = x0[0];
    v.zw = x1[0];
    *reinterpret_cast<cf2*>(dst) = v;
    v.xy = tw<+1>(x0[1], w[0][0]);
    v.zw = tw8_second<+1, W1H, 1>(x1[1], w);
    *reinterpret_cast<cf2*>(dst + Lay8::SA * 1) = v;
    v.xy = tw<+1>(x0[2], w[0][1]);
    v.zw = tw8_second<+1, W1H, 2>(x1[2], w);
    *reinterpret_cast<cf2*>(dst + Lay8::SA * 2) = v;
    v.xy = tw<+1>(x0[3], w[0][2]);
    v.zw = tw8_second<+1, W1H, 3>(x1[3], w);
    *reinterpret_cast<cf2*>(dst + Lay8::SA * 3) = v;
    v.xy = tw<+1>(x0[4], w[0][3]);
    v.zw = tw8_second<+1, W1H, 4>(x1[4], w);
    *reinterpret_cast<cf2*>(dst + Lay8::SA * 4) = v;
}

// pass 2: radix 10 over b for (alpha, j'' = 10 c + d); t2 in LDS: [beta - 1][j'']
ACQ_HD void corr8_phase2(int tid, const cf* t2_8, cf* lds) {
    if (tid >= NT8) return;
    const int al = tid / 100, jpp = tid - 100 * al, cdig = jpp / 10, ddig = jpp - 10 * cdig;
    cf* p = lds + slot8(al, 0, cdig, ddig);
    cf x[R8], y[R8];
#pragma unroll
    for (int b = 0; b < R8; ++b) x[b] = p[Lay8::SB * b];
    radix10<+1>(x, y);
    p[0] = y[0];
#pragma unroll
    for (int be = 1; be < R8; ++be) p[Lay8::SB * be] = tw<+1>(y[be], t2_8[(be - 1) * 100 + jpp]);
}

// pass 3: radix 10 over c for (alpha, beta, d); t3 in LDS: [gamma - 1][d]
ACQ_HD void corr8_phase3(int tid, const cf* t3_8, cf* lds) {
    if (tid >= NT8) return;
    const int al = tid / 100, r = tid - 100 * al, be = r / 10, ddig = r - 10 * be;
    cf* p = lds + slot8(al, be, 0, ddig);
    cf x[R8], y[R8];
#pragma unroll
    for (int cdig = 0; cdig < R8; ++cdig) x[cdig] = p[Lay8::SC * cdig];
    radix10<+1>(x, y);
    p[0] = y[0];
#pragma unroll
    for (int ga = 1; ga < R8; ++ga) p[Lay8::SC * ga] = tw<+1>(y[ga], t3_8[(ga - 1) * 10 + ddig]);
}

// pass 4 + polyphase rotation + accumulation: thread tid owns rho = tid = alpha + 5 beta + 50 gamma.
// acc[m] accumulates y[n] for n = 500 m + rho; b = bq8[q][rho], wqv[m] = W_80^{q m} (wave-uniform).
template <int MC>
ACQ_HD void corr8_phase4(int tid, cf b, const cf* wqv, const cf* lds, cf* acc) {
    if (tid >= NT8) return;
    const int ga = tid / 50, r = tid - 50 * ga, be = r / 5, al = r - 5 * be;
    const cf* p = lds + slot8(al, be, ga, 0);
    cf x[R8], y[R8];
#pragma unroll
    for (int d = 0; d < R8; d += 2) ld2(p + d, x[d], x[d + 1]);
    radix10<+1>(x, y);
#pragma unroll
    for (int d = 0; d < R8; ++d) y[d] = cmulc(y[d], b);
#pragma unroll
    for (int m = 0; m < MC; ++m) acc[m] = cmacc_u(acc[m], y[m % R8], wqv[m]);
}

// peak scan over the first S lags (:190-194), this thread's share (lags 500 m + tid), ascending
template <int MC>
ACQ_HD void corr8_scan(int tid, int S, const cf* acc, float& mx, int& mi, float& sum) {
    mx = 0.f;
    mi = 0;
    sum = 0.f;
    if (tid >= NT8) return;
#pragma unroll
    for (int m = 0; m < MC; ++m) {
        const int n = NT8 * m + tid;
        const float p = (n < S) ? acc[m].x * acc[m].x + acc[m].y * acc[m].y : 0.f;
        const bool up = p > mx;
        mx = up ? p : mx;
        mi = up ? n : mi;
        sum += p;
    }
}

}  // namespace acq
