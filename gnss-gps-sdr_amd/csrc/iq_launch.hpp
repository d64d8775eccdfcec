// iq_launch.hpp -- argument block and launchers of iq_kernels.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "iq_convert.hpp"

namespace acq {

struct IqArgs {
    const uint8_t* iq;   // interleaved I,Q bytes, 2 per sample (16-byte aligned)
    uint8_t* bits;       // ceil(n_samples / 8) output bytes, LSB first
    size_t n_samples;
    size_t first_sample; // capture sample index of iq[0] (the mixer's n)
    IqConv conv;         // format, mean, mixer (iq_convert.hpp)
};

void launch_iq_sums(const uint8_t* iq, size_t n_samples, int is_signed, unsigned long long* sums, hipStream_t s);
void launch_iq_to_bits(const IqArgs& a, hipStream_t s);
void launch_iq_to_real(const IqArgs& a, float* out, hipStream_t s);  // a.bits unused

}  // namespace acq
