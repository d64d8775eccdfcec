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

// ---------------------------------------------------------------------------------------
// The reference's own test signal, gps_sig_gen.m:8-41 (the script that wrote gps_sig_tmp.bin): one PRN, noise-free, BPSK
// with navigation bits of 20 code periods, 8 samples per chip, raised-cosine shaping (MATLAB rcosine(1, 8): roll-off 0.5,
// +-3 chips, 49 taps), carrier at a quarter of the sampling rate, 1 bit per sample:
//   s[m] = sum_j d[j] h[m - 8 j]      (conv(), accumulated oldest chip first like MATLAB's filter, plain double adds)
//   y[m] = s[m] cos(((2 pi fc) m) (1 / ca_rate))      bit = 1 if y < 0 or y == 0 (fwrite rounds the script's 0.5 up)
// Double precision and the script's own operation order, because on odd m the cosine is the ~1e-10 rounding residue of
// its argument and in 287 900 samples of the bundled file the shaped pulses cancel to +-1e-17: both signs are decided by
// rounding.  With the file's 100 navigation bits this reproduces gps_sig_tmp.bin bit for bit (tests/test_siggen.py).
// The taps are rcosine(1, 8) as IEEE doubles (h(t) = sinc(t) cos(pi t / 2) / (1 - t^2), t = k/8 - 3; t = +-1 -> sin(pi) / 4).
__constant__ double c_rc[49] = {
    0x1.2972f529d570dp-110, 0x1.2a3b74d882f6ap-10, 0x1.38ca36608bcdap-8, 0x1.5a3aace0dc09dp-7,
    0x1.18f7a0110173ep-6, 0x1.6b7d46d0781cep-6, 0x1.74baeb06b1e48p-6, 0x1.06033a3318282p-6,
    -0x1.df63f92f267c1p-57, -0x1.9efd294c27da5p-6, -0x1.d7f6a7ef8ad01p-5, -0x1.77ac1861056bfp-4,
    -0x1.ebb1581dc28abp-4, -0x1.113c3cf2ada5ep-3, -0x1.f5c461e58aedbp-4, -0x1.45bc139efcd69p-4,
    0x1.1a62633145c07p-55, 0x1.daa4571ade237p-4, 0x1.0ccdc6baf8240p-2, 0x1.b7473e8a172cap-2,
    0x1.334ed7129996cp-1, 0x1.847ab07b99f87p-1, 0x1.c643ce7028ce9p-1, 0x1.f11f44233a675p-1,
    0x1.0000000000000p+0, 0x1.f11f44233a675p-1, 0x1.c643ce7028ce9p-1, 0x1.847ab07b99f87p-1,
    0x1.334ed7129996cp-1, 0x1.b7473e8a172cap-2, 0x1.0ccdc6baf8240p-2, 0x1.daa4571ade237p-4,
    0x1.1a62633145c07p-55, -0x1.45bc139efcd69p-4, -0x1.f5c461e58aedbp-4, -0x1.113c3cf2ada5ep-3,
    -0x1.ebb1581dc28abp-4, -0x1.77ac1861056bfp-4, -0x1.d7f6a7ef8ad01p-5, -0x1.9efd294c27da5p-6,
    -0x1.df63f92f267c1p-57, 0x1.06033a3318282p-6, 0x1.74baeb06b1e48p-6, 0x1.6b7d46d0781cep-6,
    0x1.18f7a0110173ep-6, 0x1.5a3aace0dc09dp-7, 0x1.38ca36608bcdap-8, 0x1.2a3b74d882f6ap-10,
    0x1.2972f529d570dp-110};

__global__ __launch_bounds__(256) void k_siggen(SigArgs a) {
    const size_t byte = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (byte >= a.n_bytes) return;
    const long long n_chip = (long long)a.n_data * 20 * 1023;
    unsigned out = 0;
    for (int k = 0; k < 8; ++k) {
        const long long m = (long long)byte * 8 + k;
        if (m >= a.n_samples) break;
        const long long jmax = m >> 3;
        double acc = 0.0;
#pragma unroll
        for (int o = 6; o >= 0; --o) {  // oldest contributing chip first
            const long long j = jmax - o;
            const int tap = (int)(m - 8 * j);
            if (j >= 0 && j < n_chip && tap <= 48) {
                const int idx = (int)(j % 1023);
                const int chip = ((c_chips[a.sv][idx >> 5] >> (idx & 31)) & 1u) ? -1 : 1;  // 1 - 2 c
                const double d = (double)(chip * (int)a.data[j / (20 * 1023)]);
                acc = __dadd_rn(acc, __dmul_rn(d, c_rc[tap]));
            }
        }
        const double x = __dmul_rn(__dmul_rn(a.two_pi_fc, (double)m), a.inv_rate);
        const double y = __dmul_rn(acc, cos(x));
        out |= ((y < 0.0 || y == 0.0) ? 1u : 0u) << k;
    }
    a.bits[byte] = (uint8_t)out;
}
// gps_sig_gen.m:21-30, the script's other output (gps_sig_tmp_for_hackrf_tx.bin, replayed through a HackRF in README.md
// section 2.2): x = conv(rcosine(1, 8), [data x n_repeat]) .* 50 as I, zeros as Q, interleaved int8 (fwrite rounds to nearest,
// ties away from zero, and saturates) -- the same shaped baseband as k_siggen's, at IF 0.  One complex sample per thread.
__global__ __launch_bounds__(256) void k_siggen_tx(SigTxArgs a) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n_samples) return;
    const long long m = a.first_sample + (long long)i;
    const long long per_rep = (long long)a.n_data * 20 * 1023, n_chip = per_rep * a.n_repeat;
    const long long jmax = m >> 3;
    double acc = 0.0;
#pragma unroll
    for (int o = 6; o >= 0; --o) {  // oldest contributing chip first
        const long long j = jmax - o;
        const int tap = (int)(m - 8 * j);
        if (j >= 0 && j < n_chip && tap <= 48) {
            const int idx = (int)(j % 1023);
            const int chip = ((c_chips[a.sv][idx >> 5] >> (idx & 31)) & 1u) ? -1 : 1;
            const double d = (double)(chip * (int)a.data[(j % per_rep) / (20 * 1023)]);
            acc = __dadd_rn(acc, __dmul_rn(d, c_rc[tap]));
        }
    }
    double v = round(__dmul_rn(acc, 50.0));  // C round(): half away from zero, like MATLAB's
    v = v > 127.0 ? 127.0 : (v < -128.0 ? -128.0 : v);
    char2 o2;
    o2.x = (signed char)(int)v;
    o2.y = 0;
    reinterpret_cast<char2*>(a.iq)[i] = o2;
}
void launch_siggen_tx(const SigTxArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(k_siggen_tx, dim3((unsigned)((a.n_samples + 255) / 256)), dim3(256), 0, s, a);
}
void launch_siggen(const SigArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(k_siggen, dim3((unsigned)((a.n_bytes + 255) / 256)), dim3(256), 0, s, a);
}

void launch_generate(const GenArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(k_generate, dim3((unsigned)((a.n_bytes + 255) / 256)), dim3(256), 0, s, a);
}

}  // namespace acq
