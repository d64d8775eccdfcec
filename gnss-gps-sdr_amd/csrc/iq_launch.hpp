// iq_launch.hpp -- argument block and launchers of iq_kernels.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace acq {

struct IqArgs {
    const uint8_t* iq;   // interleaved I,Q bytes, 2 per sample (16-byte aligned)
    uint8_t* bits;       // ceil(n_samples / 8) output bytes, LSB first
    size_t n_samples;
    int is_signed;       // 0: uint8 offset 128 (rtl-sdr), 1: int8 (HackRF)
    int mix;             // 0: real part only, 1: real(y * exp(i theta))
    double mean_i, mean_q;
    double two_pi_fc;    // (2*pi)*fc
    double inv_fs;       // 1/fs
};

void launch_iq_sums(const uint8_t* iq, size_t n_samples, int is_signed, unsigned long long* sums, hipStream_t s);
void launch_iq_to_bits(const IqArgs& a, hipStream_t s);

}  // namespace acq
