// pk_fma_stream -- what the fp32 vector pipe of THIS box sustains on a pure v_pk_fma_f32 stream at k_corr's residency:
// 256-thread workgroups, three per CU (44 KB of LDS each pins that) = three waves per SIMD, no memory traffic, for a few
// seconds, so that the clock is the one the power cap allows under load.  bench.py runs it in its untimed part and reports
// the rate as roofline.pk_fma_stream_TF next to the 157.3 TFLOP/s of the datasheet (2.4 GHz x 4 cycles per packed
// instruction): the ceiling a kernel made of nothing but packed FMAs would reach here (measurement aid, not product code).
// Two operand forms: three distinct VGPR pairs (what most of k_corr's FMAs look like) and two.
// usage: pk_fma_stream [seconds]   ->  one JSON line on stdout
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>

typedef float cf __attribute__((ext_vector_type(2)));
#define REP 32

template <int OP> __global__ __launch_bounds__(256, 3) void k_stream(float* out, int iters) {
    __shared__ float pad[11000];  // 44 KB: three workgroups per CU, like k_corr
    cf p[16];
    for (int i = 0; i < 16; i++) {
        p[i].x = threadIdx.x * 1e-6f + i * 1e-3f;  // |values| < 1 and multipliers < 1: the recurrences stay bounded
        p[i].y = threadIdx.x * 2e-6f - i * 1e-3f;
    }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < REP; r++) {
            const int i = r & 7, j = 8 + ((r + 3) & 7), l = 8 + ((r + 5) & 7);
            if (OP == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "v"(p[j]), "v"(p[l]));
            if (OP == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p[i]) : "v"(p[j]));
        }
    }
    float acc = 0.f;
    for (int i = 0; i < 16; i++) acc += p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc + pad[threadIdx.x];
}

template <int OP> static double run(float* d, int grid, double seconds, double* ns_per_instr) {
    const int iters = 20000;  // ~15 ms per launch
    hipLaunchKernelGGL(k_stream<OP>, dim3(grid), dim3(256), 0, 0, d, 100);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    double total_ms = 0;
    long launches = 0;
    const auto t0 = std::chrono::steady_clock::now();
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
        (void)hipEventRecord(e0);
        for (int k = 0; k < 8; ++k) hipLaunchKernelGGL(k_stream<OP>, dim3(grid), dim3(256), 0, 0, d, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        total_ms += ms;
        launches += 8;
    }
    const double instr_per_wave = (double)launches * iters * REP;  // packed FMAs issued by every wave
    const double flops = instr_per_wave * 64.0 * 4.0 * (grid * 4.0);  // 64 lanes x 2 FMAs x 2 flops, grid x 4 waves
    *ns_per_instr = total_ms * 1e6 / (instr_per_wave * 3.0);  // per wave-instruction per SIMD at three waves per SIMD
    return flops / (total_ms * 1e-3) / 1e12;
}

int main(int argc, char** argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 3.0;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) {
        fprintf(stderr, "pk_fma_stream: no HIP device\n");
        return 2;
    }
    const int cus = prop.multiProcessorCount, grid = cus * 3;
    float* d = nullptr;
    if (hipMalloc((void**)&d, (size_t)grid * 256 * sizeof(float)) != hipSuccess) return 2;
    double ns3 = 0, ns2 = 0;
    const double tf3 = run<0>(d, grid, seconds * 0.75, &ns3);
    const double tf2 = run<1>(d, grid, seconds * 0.25, &ns2);
    printf("{\"pk_fma_stream_TF\": %.3f, \"ns_per_wave_instr_per_simd\": %.4f, \"pk_fma_2reg_stream_TF\": %.3f, \"ns_per_wave_instr_per_simd_2reg\": %.4f, "
           "\"compute_units\": %d, \"waves_per_simd\": 3, \"seconds\": %.2f, \"form\": \"v_pk_fma_f32 v[a], v[b], v[c], v[a] (three distinct VGPR pairs), 8 independent chains per wave\"}\n",
           tf3, ns3, tf2, ns2, cus, seconds);
    (void)hipFree(d);
    return 0;
}
