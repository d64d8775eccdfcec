// iq_kernels.hip -- 8-bit IQ capture -> 1-bit real-IF stream, on the device.
//
// The reference does this offline in MATLAB before gps_test ever runs:
//   proc_rtl_bin_for_gps.m:12-26,31-47   rtl-sdr  uint8 I,Q (offset 128): y = y-128; y = I + jQ;
//                                        y = y - mean(y); [y = real(y .* exp(1i*2*pi*fc*n/fs))];
//                                        y = (1-sign(y))/2; fwrite(..., 'ubit1')
//   proc_hackrf_bin_for_gps.m:7-19       HackRF int8 I,Q: same without the offset
// i.e. DC removal over the whole capture, optional mix to a real IF, sign, LSB-first packing --
// exactly the byte stream Sample() (c/search_offline.cpp:136-146) unpacks.  HBM-bound byte
// work: two streaming passes (integer sums for the mean; convert + pack), 16-byte loads, one
// output byte per thread.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "iq_launch.hpp"

namespace acq {

// pass 1: exact integer sums of I and Q (MATLAB's mean() of these integer-valued doubles is exact too)
__global__ __launch_bounds__(256) void k_iq_sums(const uint8_t* __restrict__ iq, size_t n_samples, int is_signed,
                                                 unsigned long long* sums /* [2] biased sums */) {
    long long si = 0, sq = 0;
    const size_t n16 = n_samples / 8;  // 16-byte groups of 8 samples
    const uint4* p = reinterpret_cast<const uint4*>(iq);
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < n16; g += (size_t)gridDim.x * blockDim.x) {
        const uint4 v = p[g];
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (is_signed) {
                si += (int8_t)(w[k] & 0xff) + (int8_t)((w[k] >> 16) & 0xff);
                sq += (int8_t)((w[k] >> 8) & 0xff) + (int8_t)((w[k] >> 24) & 0xff);
            } else {
                si += (int)(w[k] & 0xff) + (int)((w[k] >> 16) & 0xff) - 256;
                sq += (int)((w[k] >> 8) & 0xff) + (int)((w[k] >> 24) & 0xff) - 256;
            }
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {  // tail (< 8 samples)
        for (size_t s = n16 * 8; s < n_samples; ++s) {
            si += is_signed ? (int)(int8_t)iq[2 * s] : (int)iq[2 * s] - 128;
            sq += is_signed ? (int)(int8_t)iq[2 * s + 1] : (int)iq[2 * s + 1] - 128;
        }
    }
    // wave reduction, then one atomic pair per wave
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        si += __shfl_down(si, off, 64);
        sq += __shfl_down(sq, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&sums[0], (unsigned long long)si);
        atomicAdd(&sums[1], (unsigned long long)sq);
    }
}

// pass 2: one output byte (8 samples) per thread
__global__ __launch_bounds__(256) void k_iq_to_bits(IqArgs a) {
    const size_t byte = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t n_bytes = (a.n_samples + 7) / 8;
    if (byte >= n_bytes) return;
    const size_t s0 = byte * 8;
    unsigned raw[4] = {0, 0, 0, 0};
    if (s0 + 8 <= a.n_samples) {
        const uint4 v = reinterpret_cast<const uint4*>(a.iq)[byte];
        raw[0] = v.x; raw[1] = v.y; raw[2] = v.z; raw[3] = v.w;
    } else {
        for (size_t s = s0; s < a.n_samples; ++s) {
            const unsigned pair = a.iq[2 * s] | ((unsigned)a.iq[2 * s + 1] << 8);
            raw[(s - s0) >> 1] |= pair << (16 * ((s - s0) & 1));
        }
    }
    const unsigned out = iq8_byte(raw, a.first_sample + s0, a.first_sample + a.n_samples, a.conv);
    a.bits[byte] = (uint8_t)out;
}

// multi-bit path (SURVEY.md section 8f.1 "direct float path"; no reference counterpart: gps_test only reads 1-bit files): the
// same real-IF value kept as a float instead of its sign, with the quadrature LO of Sample() (:143-153) applied as signs
// (I = +-x by lo_cos, Q = +-x by lo_sin; mask bit 1 <-> factor -1, like Bipolar(bit ^ lo); the LO phase restarts with every
// block, :131).  One thread per 8 samples; out[block][40000] complex.
// sub > 1 (Doppler grid finer than a bin, gpsacq_set_doppler_step): every block is written sub times, copy r turned by
// exp(-2 pi i (r / sub) n / 40000) -- the carrier offset of r / sub bins that oracle_sample_ramped() applies to the mixed samples
// (ramp in double, product rounded to float) -- in the order [block][r] of the block spectra.
__device__ __forceinline__ float2 sub_bin_turn(double re, double im, int r, int sub, int n) {
    if (r != 0) {
        const double th = 6.283185307179586476925286766559 * ((double)r / (double)sub) * (double)n / 40000.0;
        double sn, cs;
        sincos(th, &sn, &cs);
        const double a = re * cs + im * sn, b = im * cs - re * sn;
        re = a;
        im = b;
    }
    return make_float2((float)re, (float)im);
}

__global__ __launch_bounds__(256) void k_iq_to_mixed(IqArgs a, size_t stride_samples, size_t n_blocks, int sub, const uint8_t* __restrict__ cos_mask,
                                                     const uint8_t* __restrict__ sin_mask, float2* __restrict__ out) {
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // group of 8 samples inside a copy of a block: 5000 groups each
    const size_t item = g / 5000, blk = item / (size_t)sub;
    const int grp = (int)(g - item * 5000), r = (int)(item - blk * (size_t)sub);
    if (blk >= n_blocks) return;
    const size_t s0 = blk * stride_samples + (size_t)grp * 8;  // sample index inside the batch (stride_samples is a multiple of 8)
    unsigned raw[4] = {0, 0, 0, 0};
    if (s0 + 8 <= a.n_samples) {
        const uint4 v = reinterpret_cast<const uint4*>(a.iq)[s0 / 8];
        raw[0] = v.x; raw[1] = v.y; raw[2] = v.z; raw[3] = v.w;
    }
    const unsigned cm = cos_mask[grp], sm = sin_mask[grp];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float v = (s0 + k < a.n_samples) ? (float)iq8_value((raw[k >> 1] >> (16 * (k & 1))) & 0xffffu, a.first_sample + s0 + k, a.conv) : 0.f;
        out[item * 40000 + (size_t)grp * 8 + k] = sub_bin_turn(((cm >> k) & 1u) ? -(double)v : (double)v, ((sm >> k) & 1u) ? -(double)v : (double)v, r, sub, grp * 8 + k);
    }
}
// complex-baseband path: the capture already is what Sample() builds in fwd_buf -- I + jQ at IF 0, e.g. the int8 +-30 file the
// reference's own converter writes for HackRF replay (c/conv_1bit_bin_to_hackrf_bin.cpp:61-80: I = Bipolar(bit ^ lo_sin),
// Q = Bipolar(bit ^ lo_cos), the two components of fwd_buf, :149-150 of search_offline.cpp) -- so no LO here: the samples
// (less their mean, turned by exp(i theta) when a residual IF is named) go to the forward transform as they are.
__global__ __launch_bounds__(256) void k_iq_to_complex(IqArgs a, size_t stride_samples, size_t n_blocks, int sub, float2* __restrict__ out) {
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t item = g / 5000, blk = item / (size_t)sub;
    const int grp = (int)(g - item * 5000), r = (int)(item - blk * (size_t)sub);
    if (blk >= n_blocks) return;
    const size_t s0 = blk * stride_samples + (size_t)grp * 8;
    unsigned raw[4] = {0, 0, 0, 0};
    if (s0 + 8 <= a.n_samples) {
        const uint4 v = reinterpret_cast<const uint4*>(a.iq)[s0 / 8];
        raw[0] = v.x; raw[1] = v.y; raw[2] = v.z; raw[3] = v.w;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        double re = 0.0, im = 0.0;
        if (s0 + k < a.n_samples) iq8_complex((raw[k >> 1] >> (16 * (k & 1))) & 0xffffu, a.first_sample + s0 + k, a.conv, re, im);
        // the sample as the float buffer of the whole-bin path holds it, then the turn
        out[item * 40000 + (size_t)grp * 8 + k] = sub_bin_turn((double)(float)re, (double)(float)im, r, sub, grp * 8 + k);
    }
}
void launch_iq_to_mixed(const IqArgs& a, size_t stride_samples, size_t n_blocks, int sub, const uint8_t* cos_mask, const uint8_t* sin_mask, void* out, hipStream_t s) {
    const size_t groups = n_blocks * (size_t)sub * 5000;
    hipLaunchKernelGGL(k_iq_to_mixed, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, s, a, stride_samples, n_blocks, sub, cos_mask, sin_mask, (float2*)out);
}
void launch_iq_to_complex(const IqArgs& a, size_t stride_samples, size_t n_blocks, int sub, void* out, hipStream_t s) {
    const size_t groups = n_blocks * (size_t)sub * 5000;
    hipLaunchKernelGGL(k_iq_to_complex, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, s, a, stride_samples, n_blocks, sub, (float2*)out);
}

void launch_iq_sums(const uint8_t* iq, size_t n_samples, int is_signed, unsigned long long* sums, hipStream_t s) {
    hipLaunchKernelGGL(k_iq_sums, dim3(2048), dim3(256), 0, s, iq, n_samples, is_signed, sums);
}
void launch_iq_to_bits(const IqArgs& a, hipStream_t s) {
    const size_t n_bytes = (a.n_samples + 7) / 8;
    hipLaunchKernelGGL(k_iq_to_bits, dim3((unsigned)((n_bytes + 255) / 256)), dim3(256), 0, s, a);
}

}  // namespace acq
