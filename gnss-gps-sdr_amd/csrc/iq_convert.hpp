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

// 8 consecutive samples (16 bytes: I0 Q0 I1 Q1 ...) starting at capture sample n0 -> one byte of the 1-bit stream
// (sample n0 + k in bit k).  Samples n >= n_samples (ragged tail) give 0 bits.
__device__ __forceinline__ unsigned iq8_byte(const unsigned (&raw)[4], size_t n0, size_t n_samples, const IqConv& c) {
    unsigned out = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const size_t n = n0 + k;
        const unsigned pair = (raw[k >> 1] >> (16 * (k & 1))) & 0xffffu;
        const double r = iq8_value(pair, n, c);
        // (1 - sign(r)) / 2 written as ubit1: r > 0 -> 0, r < 0 -> 1, r == 0 -> 0.5 which fwrite rounds to 1
        const unsigned bit = (n < n_samples) ? (r > 0.0 ? 0u : 1u) : 0u;
        out |= bit << k;
    }
    return out;
}

}  // namespace acq
