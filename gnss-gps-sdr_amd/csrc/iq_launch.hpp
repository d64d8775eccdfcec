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
// multi-bit path: out[block][40000] complex floats, LO applied as signs (a.bits unused)
void launch_iq_to_mixed(const IqArgs& a, size_t stride_samples, size_t n_blocks, int sub, const uint8_t* cos_mask, const uint8_t* sin_mask, void* out, hipStream_t s);
void launch_iq_to_complex(const IqArgs& a, size_t stride_samples, size_t n_blocks, int sub, void* out, hipStream_t s);

}  // namespace acq
