// Micro-benchmark: LDS cycles per wave-instruction of the 8- and 16-byte DS reads/writes on gfx950,
// with the address patterns of k_corr's three passes (lane stride in 8-byte elements).
//   hipcc --offload-arch=gfx950 -O3 lds_rate.hip -o lds_rate && ./lds_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
#define REP 16
// OP: 0 ds_read_b64 x2, 1 ds_read2_b64, 2 ds_read_b128, 3 ds_write_b64 x2, 4 ds_write2_b64, 5 ds_write_b128
template <int OP> __global__ __launch_bounds__(256) void k(float* out, int iters, int lane_stride, int off2) {
    __shared__ __attribute__((aligned(16))) f2 lds[5600];
    for (int i = threadIdx.x; i < 5600; i += 256) lds[i] = f2{(float)i, 1.f};
    __syncthreads();
    const unsigned a = (unsigned)(size_t)(__attribute__((address_space(3))) f2*)lds + 8u * (unsigned)((threadIdx.x * lane_stride) % 2500);
    f2 r0 = f2{0, 0}, r1 = f2{0, 0};
    f4 q = f4{0, 0, 0, 0};
    float acc = 0;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < REP; r++) {
            if (OP == 0) asm volatile("ds_read_b64 %0, %2 offset:%3\n\tds_read_b64 %1, %2 offset:%4" : "=v"(r0), "=v"(r1) : "v"(a), "n"(r * 16), "n"(r * 16 + 200));
            if (OP == 1) asm volatile("ds_read2_b64 %0, %1 offset0:%2 offset1:%3" : "=v"(q) : "v"(a), "n"(r * 2), "n"(r * 2 + 25));
            if (OP == 2) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q) : "v"(a), "n"(r * 16));
            if (OP == 3) asm volatile("ds_write_b64 %1, %0 offset:%2\n\tds_write_b64 %1, %0 offset:%3" ::"v"(r0), "v"(a), "n"(r * 16), "n"(r * 16 + 200));
            if (OP == 4) asm volatile("ds_write2_b64 %1, %0, %0 offset0:%2 offset1:%3" ::"v"(r0), "v"(a), "n"(r * 2), "n"(r * 2 + 25));
            if (OP == 5) asm volatile("ds_write_b128 %1, %0 offset:%2" ::"v"(q), "v"(a), "n"(r * 16));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r0), "+v"(r1), "+v"(q));
        acc += r0.x + r1.x + q.x;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc + (float)off2;
}
template <int OP> void run(const char* name, float* d, int wgs_per_cu, int lane_stride) {
    const int iters = 2000, grid = 256 * wgs_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(256), 0, 0, d, 10, lane_stride, 0);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(256), 0, 0, d, iters, lane_stride, 0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    // 16 bytes per lane per "op": cycles per CU per 16-byte wave-op (4 waves per WG share one LDS)
    const double ops_per_cu = (double)iters * REP * 4 * wgs_per_cu;
    const double cyc = ms * 1e-3 * 2.4e9 / ops_per_cu;
    printf("%-16s stride=%2d WG/CU=%d  %.3f ms  %.2f cycles per 16 B/lane wave-op per CU -> %.0f B/clk/CU\n", name, lane_stride, wgs_per_cu, ms, cyc,
           1024.0 / cyc);
}
int main() {
    float* d;
    hipMalloc(&d, 256 * 8 * 256 * 4);
    for (int st : {1, 2, 25}) {
        for (int w : {1, 3}) {
            run<0>("2x ds_read_b64", d, w, st);
            run<1>("ds_read2_b64", d, w, st);
            if (st % 2 == 0) run<2>("ds_read_b128", d, w, st);
            run<3>("2x ds_write_b64", d, w, st);
            run<4>("ds_write2_b64", d, w, st);
            if (st % 2 == 0) run<5>("ds_write_b128", d, w, st);
        }
    }
    return 0;
}
