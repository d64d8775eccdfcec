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

// gps_sig_gen.m's own signal (gen_kernels.hip, k_siggen)
struct SigArgs {
    uint8_t* bits;       // [n_bytes] output, LSB first
    size_t n_bytes;
    long long n_samples; // n_data * 20 * 1023 * 8 + 48 (the tail of the shaping filter); bits beyond it are 0
    const int8_t* data;  // [n_data] navigation bits +-1 (device)
    int n_data;
    int sv;              // PRN index 0..31
    double two_pi_fc;    // (2 pi) * (ca_rate / 4), rounded like the script's left-to-right product
    double inv_rate;     // 1 / ca_rate
};
// gps_sig_gen.m:21-30, the HackRF transmit file (gen_kernels.hip, k_siggen_tx)
struct SigTxArgs {
    int8_t* iq;            // [2 * n_samples] interleaved I, Q (Q = 0)
    size_t n_samples;      // complex samples to write
    long long first_sample;// index of the first one in the stream of n_repeat * n_data * 20 * 1023 * 8 + 48
    const int8_t* data;    // [n_data] navigation bits +-1 (device)
    int n_data, n_repeat, sv;
};
void launch_siggen_tx(const SigTxArgs& a, hipStream_t s);
hipError_t upload_chips(const uint32_t* host);
void launch_siggen(const SigArgs& a, hipStream_t s);
void launch_generate(const GenArgs& a, hipStream_t s);

}  // namespace acq
