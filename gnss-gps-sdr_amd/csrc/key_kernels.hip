// key_kernels.hip -- the small kernels of the multi-GPU merge (gpsacq_multi.cpp, include/gpsacq.h gpsacq_multi_*): a peak
// becomes a 64-bit key whose integer MAX reproduces the reference's ordering (c/search_offline.cpp:196-198), keys are
// reduced per PRN, MAX-merged between engines that share a GPU, and the winner's max_pwr is carried separately.
// Kept apart from acq_kernels.hip so that the hash that ties profiles/traffic.json to the timed kernels (bench.py
// kernel_source_sha) only moves when those kernels do.
#include <hip/hip_runtime.h>

#include "../../include/gpsacq.h"
#include "acq_launch.hpp"

namespace acq {

// Multi-GPU merge key of a peak: integer MAX over
//   key = snr bits << 32 | (0xFFFF - (lo_shift + kmax)) << 16 | ca_shift
// picks the higher SNR and, on equal SNR, the LOWER Doppler grid point -- the reference's strict '>' scan over
// ascending dop (:196-198).  Non-negative IEEE floats order like their bit patterns.
__device__ __forceinline__ unsigned long long peak_key(const Peak p, int kmax) {
    const unsigned long long snr = (unsigned long long)__float_as_uint(p.snr > 0.f ? p.snr : 0.f);
    const unsigned long long lo = (unsigned long long)(0xFFFF - (p.lo_shift + kmax)) & 0xFFFFull;
    return (snr << 32) | (lo << 16) | ((unsigned long long)p.ca_shift & 0xFFFFull);
}
__global__ void k_pack_keys(const Peak* peaks, unsigned long long* keys, int n, int kmax) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    keys[i] = peak_key(peaks[i], kmax);
}

// gpsacq_peak_keys_device(per_prn): peaks -> the 32 per-PRN best keys in ONE launch (reference schedule, task t <-> PRN t % 32):
// what a rank of the block decomposition hands to the all-reduce (bench.py's step = one search + this + one all-reduce).
// One workgroup of 1024; thread (r, sv) strides over the runs; n_tasks == 0 leaves 32 zero keys (neutral for MAX).
constexpr int KEYS_WG = 1024;  // 32 PRNs x 32 runs in flight: the kernel sits on the engine's stream between two searches
__global__ __launch_bounds__(KEYS_WG) void k_prn_keys(const Peak* peaks, int n_tasks, int kmax, unsigned long long* best) {
    __shared__ unsigned long long part[KEYS_WG];
    const int sv = threadIdx.x & 31, lane_run = threadIdx.x >> 5;
    unsigned long long k = 0;
    for (int t = lane_run * 32 + sv; t < n_tasks; t += KEYS_WG) {
        const unsigned long long kt = peak_key(peaks[t], kmax);
        k = kt > k ? kt : k;
    }
    part[threadIdx.x] = k;
    __syncthreads();
    if (threadIdx.x < 32) {
        for (int r = 1; r < KEYS_WG / 32; ++r) k = part[r * 32 + sv] > k ? part[r * 32 + sv] : k;
        best[sv] = k;
    }
}

// Reference schedule (task t <-> PRN t % 32): the best key of each PRN over all runs of this device -- what the one
// all-reduce of the block decomposition carries (32 keys).  One workgroup, thread (r, sv) strides over the runs.
// best_pwr[sv]: max_pwr of the peak that made PRN sv's best key (what the key itself has no room for).
__global__ __launch_bounds__(WG) void k_prn_best(const unsigned long long* keys, const Peak* peaks, int n_tasks, unsigned long long* best, float* best_pwr) {
    __shared__ unsigned long long part[WG];
    __shared__ int part_t[WG];
    const int sv = threadIdx.x & 31, lane_run = threadIdx.x >> 5;  // 8 runs in flight per pass
    unsigned long long k = 0;
    int kt = -1;
    for (int t = lane_run * 32 + sv; t < n_tasks; t += WG)
        if (keys[t] > k) {
            k = keys[t];
            kt = t;
        }
    part[threadIdx.x] = k;
    part_t[threadIdx.x] = kt;
    __syncthreads();
    if (threadIdx.x < 32) {
        for (int r = 1; r < WG / 32; ++r)
            if (part[r * 32 + sv] > k) {
                k = part[r * 32 + sv];
                kt = part_t[r * 32 + sv];
            }
        best[sv] = k;
        best_pwr[sv] = kt >= 0 ? peaks[kt].max_pwr : 0.f;
    }
}

// Engines that share one physical GPU merge their keys on the device (RCCL joins distinct GPUs only): dst = max(dst, src).
__global__ void k_max_u64(unsigned long long* dst, const unsigned long long* src, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && src[i] > dst[i]) dst[i] = src[i];
}
__global__ void k_max_f32(float* dst, const float* src, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && src[i] > dst[i]) dst[i] = src[i];
}
// after the merge: the engine whose own key won reports its peak's max_pwr, everybody else 0 (then MAX-merged like the keys)
__global__ void k_winner_pwr(const unsigned long long* own, const unsigned long long* merged, const float* own_pwr, float* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (own[i] != 0 && own[i] == merged[i]) ? own_pwr[i] : 0.f;
}
__global__ void k_peak_pwr(const Peak* peaks, float* pwr, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) pwr[i] = peaks[i].max_pwr;
}

// gpsacq_cycle_stamp_device: the shader-cycle counter (s_memtime) of every compute unit, written to out[__smid()] -- slot =
// XCC_ID << 6 | SE_ID << 4 | CU_ID (512 slots).  The counter is PER COMPUTE UNIT: each CU's starts from its own offset (two CUs
// of one XCD read values tens of millions of ticks apart) and stands still while the CU is clock-gated, so only two readings of
// the SAME CU make a difference that means anything.  4096 single-wave workgroups reach every CU of an idle or a busy GPU; the
// waves that land on one CU write the same counter a few cycles apart.  Two calls around a stretch of work that keeps the CUs
// busy, divided by the time between them (HIP events at the same two points of the stream), give the average clock each CU
// held in between -- measured by the GPU, no firmware averaging window in the way.
__global__ void k_cycle_stamp(unsigned long long* out) {
    if (threadIdx.x == 0) out[__smid() & (GPSACQ_STAMP_SLOTS - 1)] = __builtin_amdgcn_s_memtime();
}

// launchers (host)
void launch_cycle_stamp(unsigned long long* out, hipStream_t s) { hipLaunchKernelGGL(k_cycle_stamp, dim3(4096), dim3(64), 0, s, out); }
void launch_pack_keys(const Peak* peaks, unsigned long long* keys, int n, int kmax, hipStream_t s) {
    hipLaunchKernelGGL(k_pack_keys, dim3((n + 255) / 256), dim3(256), 0, s, peaks, keys, n, kmax);
}
void launch_prn_keys(const Peak* peaks, int n_tasks, int kmax, unsigned long long* best, hipStream_t s) {
    hipLaunchKernelGGL(k_prn_keys, dim3(1), dim3(KEYS_WG), 0, s, peaks, n_tasks, kmax, best);
}
void launch_prn_best(const unsigned long long* keys, const Peak* peaks, int n_tasks, unsigned long long* best, float* best_pwr, hipStream_t s) {
    hipLaunchKernelGGL(k_prn_best, dim3(1), dim3(WG), 0, s, keys, peaks, n_tasks, best, best_pwr);
}
void launch_max_u64(unsigned long long* dst, const unsigned long long* src, int n, hipStream_t s) {
    hipLaunchKernelGGL(k_max_u64, dim3((n + 255) / 256), dim3(256), 0, s, dst, src, n);
}
void launch_max_f32(float* dst, const float* src, int n, hipStream_t s) {
    hipLaunchKernelGGL(k_max_f32, dim3((n + 255) / 256), dim3(256), 0, s, dst, src, n);
}
void launch_winner_pwr(const unsigned long long* own, const unsigned long long* merged, const float* own_pwr, float* out, int n, hipStream_t s) {
    hipLaunchKernelGGL(k_winner_pwr, dim3((n + 255) / 256), dim3(256), 0, s, own, merged, own_pwr, out, n);
}
void launch_peak_pwr(const Peak* peaks, float* pwr, int n, hipStream_t s) {
    hipLaunchKernelGGL(k_peak_pwr, dim3((n + 255) / 256), dim3(256), 0, s, peaks, pwr, n);
}

}  // namespace acq
