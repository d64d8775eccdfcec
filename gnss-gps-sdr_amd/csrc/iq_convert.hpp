// iq_convert.hpp -- the per-sample arithmetic of the 8-bit IQ ingestion, shared by the stand-alone converter
// (iq_kernels.hip::k_iq_to_bits) and by the forward transform that reads an 8-bit IQ capture directly
// (acq_kernels.hip::k_fwd<SRC_IQ8>), so that both make the same bit from the same sample.
//
// Reference (MATLAB, run offline before gps_test): proc_rtl_bin_for_gps.m:12-26,31-47 (uint8, offset 128),
// proc_hackrf_bin_for_gps.m:7-19 (int8):  y = I + 1i*Q;  y = y - mean(y);  [y = real(y .* exp(1i*2*pi*fc*n/fs))];
// bit = (1 - sign(y)) / 2 written as 'ubit1' (0.5 rounds to 1), LSB first.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace acq {

struct IqConv {
    int is_signed;       // 0: uint8 offset 128 (rtl-sdr), 1: int8 (HackRF)
    int mix;             // 0: real part only, 1: real(y * exp(i theta))
    double mean_i, mean_q;
    double two_pi_fc;    // (2*pi)*fc
    double inv_fs;       // 1/fs
    double fc, fs;       // the same two numbers unrounded, for the HackRF script's operation order
};

// The mixer phase of capture sample n, in the operation order of the script that handles the format (MATLAB evaluates left to
// right in double): proc_rtl_bin_for_gps.m:41  1i.*2.*pi.*fc.*(0:n-1).*(1./fs)  ->  (((2 pi) fc) n) (1 / fs);
// proc_hackrf_bin_for_gps.m:14  1i.*(0:n-1).*2.6e6.*2.*pi./10e6  ->  (((n fc) 2) pi) / fs.
__device__ __forceinline__ double iq8_theta(size_t n, const IqConv& c) {
#pragma clang fp contract(off)
    if (c.is_signed) return ((((double)n * c.fc) * 2.0) * 3.141592653589793) / c.fs;
    return (c.two_pi_fc * (double)n) * c.inv_fs;
}

// one sample (I in the low byte of `pair`, Q in the high byte), capture index n -> its real-IF value before the sign
__device__ __forceinline__ double iq8_value(unsigned pair, size_t n, const IqConv& c) {
#pragma clang fp contract(off)  // MATLAB / the numpy oracle round the two products before subtracting
    double yi, yq;
    if (c.is_signed) { yi = (double)(int8_t)(pair & 0xff); yq = (double)(int8_t)(pair >> 8); }
    else { yi = (double)(int)(pair & 0xff) - 128.0; yq = (double)(int)(pair >> 8) - 128.0; }
    yi -= c.mean_i;
    yq -= c.mean_q;
    double r = yi;
    if (c.mix) {
        const double th = iq8_theta(n, c);
        double sn, cs;
        sincos(th, &sn, &cs);
        r = yi * cs - yq * sn;
    }
    return r;
}

// the same sample kept complex (the capture is at baseband already): (y - mean) * exp(i theta), theta as above
__device__ __forceinline__ void iq8_complex(unsigned pair, size_t n, const IqConv& c, double& re, double& im) {
#pragma clang fp contract(off)
    double yi, yq;
    if (c.is_signed) { yi = (double)(int8_t)(pair & 0xff); yq = (double)(int8_t)(pair >> 8); }
    else { yi = (double)(int)(pair & 0xff) - 128.0; yq = (double)(int)(pair >> 8) - 128.0; }
    yi -= c.mean_i;
    yq -= c.mean_q;
    re = yi;
    im = yq;
    if (c.mix) {
        const double th = iq8_theta(n, c);
        double sn, cs;
        sincos(th, &sn, &cs);
        re = yi * cs - yq * sn;
        im = yi * sn + yq * cs;
    }
}

// The written bit of one sample -- (1 - sign(r)) / 2 as ubit1: r > 0 -> 0, r < 0 -> 1, r == 0 -> 0.5 which fwrite rounds to 1 --
// without the double-precision sincos wherever the sign cannot depend on it (round 4: the mixer's 40 000 double sincos per block
// were the IQ forward stage's largest item).  The phase is formed exactly as iq8_theta() forms it, reduced to [-pi/4, pi/4] in
// double (two-term 2 pi and pi/2: reduction error < 1e-15 for every capture index), and sine and cosine come from float Taylor
// polynomials there (truncation < 3.1e-7 and < 2.5e-8, evaluation rounding < 5e-7): each within 1e-6 of the true value, so
// r_fast = yi cs - yq sn is within 512 x 1e-6 + 1e-4 (float products of |y| <= 256) < 7e-4 of the double result.  Only when
// |r_fast| < 5e-3 -- one sample in a few thousand -- is the bit taken from the double-precision path; elsewhere both agree on the
// sign by construction, so the stream is the same bit for bit (tests/test_iq.py, the IQ modes of tools/fuzz_gpu.py).
__device__ __forceinline__ unsigned iq8_bit(unsigned pair, size_t n, const IqConv& c) {
    if (!c.mix) return iq8_value(pair, n, c) > 0.0 ? 0u : 1u;  // no trigonometry on this path
    double yi, yq;
    if (c.is_signed) { yi = (double)(int8_t)(pair & 0xff); yq = (double)(int8_t)(pair >> 8); }
    else { yi = (double)(int)(pair & 0xff) - 128.0; yq = (double)(int)(pair >> 8) - 128.0; }
    yi -= c.mean_i;
    yq -= c.mean_q;
    const double th = iq8_theta(n, c);
    // th = k 2 pi + j pi/2 + s,  |s| <= pi/4
    const double k = rint(th * 0.15915494309189535);
    double r = fma(-k, 6.283185307179586, th);
    r = fma(-k, 2.4492935982947064e-16, r);
    const double jd = rint(r * 0.6366197723675814);
    double sd = fma(-jd, 1.5707963267948966, r);
    sd = fma(-jd, 6.123233995736766e-17, sd);
    const float sf = (float)sd, s2 = sf * sf;
    const float sn0 = sf + sf * s2 * (-0.16666667f + s2 * (8.3333333e-3f + s2 * -1.9841270e-4f));
    const float cs0 = 1.0f + s2 * (-0.5f + s2 * (4.1666667e-2f + s2 * (-1.3888889e-3f + s2 * 2.4801587e-5f)));
    const int j = (int)jd & 3;  // quadrant (two's complement: -1 & 3 = 3)
    const float sn = (j == 0) ? sn0 : (j == 1) ? cs0 : (j == 2) ? -sn0 : -cs0;
    const float cs = (j == 0) ? cs0 : (j == 1) ? -sn0 : (j == 2) ? -cs0 : sn0;
    const float rf = (float)yi * cs - (float)yq * sn;
    if (fabsf(rf) >= 5e-3f) return rf > 0.f ? 0u : 1u;
    return iq8_value(pair, n, c) > 0.0 ? 0u : 1u;
}

// 8 consecutive samples (16 bytes: I0 Q0 I1 Q1 ...) starting at capture sample n0 -> one byte of the 1-bit stream
// (sample n0 + k in bit k).  Samples n >= n_samples (ragged tail) give 0 bits.
__device__ __forceinline__ unsigned iq8_byte(const unsigned (&raw)[4], size_t n0, size_t n_samples, const IqConv& c) {
    unsigned out = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const size_t n = n0 + k;
        const unsigned pair = (raw[k >> 1] >> (16 * (k & 1))) & 0xffffu;
        const unsigned bit = (n < n_samples) ? iq8_bit(pair, n, c) : 0u;
        out |= bit << k;
    }
    return out;
}

}  // namespace acq
