// gen_kernels.hip -- synthetic 1-bit real-IF GPS L1 C/A capture generated on the device
// (the reference's counterpart is the MATLAB script gps_sig_gen.m:8-41,
// which writes one noise-free PRN at zero Doppler; here: any set of PRNs, Doppler, code phase,
// amplitude, plus white Gaussian noise, the signal model of SURVEY.md section 8d):
//   y[m] = sigma n[m] + sum_k a_k c_k[ floor((m + phi_k) CPS (1 + fd_k/L1) / fs) mod 1023 ]
//                             cos(2 pi ((fc + fd_k)/fs m + theta_k)),        bit = (y < 0)
// packed LSB first like MATLAB's 'ubit1' (gps_sig_gen.m:39-41).  One output byte per thread.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gen_launch.hpp"

namespace acq {

__constant__ uint32_t c_chips[32][32];  // 1023 chips per PRN, bit i of word i/32; 1 = chip value 1 (-> -1.0)
hipError_t upload_chips(const uint32_t* host) { return hipMemcpyToSymbol(HIP_SYMBOL(c_chips), host, sizeof(uint32_t) * 32 * 32); }

__device__ __forceinline__ uint64_t mix64(uint64_t z) {  // splitmix64 finaliser: counter-based noise
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

__global__ __launch_bounds__(256) void k_generate(GenArgs a) {
    const size_t byte = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (byte >= a.n_bytes) return;
    unsigned out = 0;
    for (int k = 0; k < 8; ++k) {
        const uint64_t m = a.first_sample + byte * 8 + k;
        // Box-Muller on two 32-bit uniforms derived from (seed, m)
        const uint64_t h = mix64(a.seed ^ (m * 0x9e3779b97f4a7c15ull));
        const float u1 = ((float)(uint32_t)(h >> 32) + 1.0f) * 2.3283064e-10f;  // (0, 1]
        const float u2 = (float)(uint32_t)h * 2.3283064e-10f;
        float y = a.noise_sigma * sqrtf(-2.0f * __logf(u1)) * __cosf(6.2831853f * u2);
        for (int s = 0; s < a.n_sats; ++s) {
            const GenSat sat = a.sats[s];
            const double r = ((double)m + sat.code_phase) * sat.chips_per_sample;
            const long long q = (long long)floor(r);
            int idx = (int)(q % 1023);
            if (idx < 0) idx += 1023;
            const float chip = ((c_chips[sat.sv][idx >> 5] >> (idx & 31)) & 1u) ? -1.0f : 1.0f;
            double ph = sat.cycles_per_sample * (double)m + sat.carrier_phase;
            ph -= floor(ph);
            y += sat.amplitude * chip * cospif(2.0f * (float)ph);
        }
        out |= (y < 0.0f ? 1u : 0u) << k;
    }
    a.bits[byte] = (uint8_t)out;
}

void launch_generate(const GenArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(k_generate, dim3((unsigned)((a.n_bytes + 255) / 256)), dim3(256), 0, s, a);
}

}  // namespace acq
