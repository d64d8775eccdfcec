// gen_launch.hpp -- argument block and launcher of gen_kernels.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace acq {

struct GenSat {
    int32_t sv;               // PRN index 0..31
    float amplitude;
    double chips_per_sample;  // CPS (1 + fd/L1) / fs
    double code_phase;        // samples
    double cycles_per_sample; // (fc + fd) / fs
    double carrier_phase;     // cycles
};
struct GenArgs {
    uint8_t* bits;
    size_t n_bytes;
    uint64_t first_sample;
    uint64_t seed;
    const GenSat* sats;  // device
    int n_sats;
    float noise_sigma;
};

hipError_t upload_chips(const uint32_t* host);
void launch_generate(const GenArgs& a, hipStream_t s);

}  // namespace acq
