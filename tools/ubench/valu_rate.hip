// Micro-benchmark: VALU issue cost of scalar vs packed fp32 ops on gfx950 (one result line per op).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
#define REP 64
template <int OP> __global__ __launch_bounds__(256) void k(float* out, int iters, float s) {
    float a[8]; f2 p[8];
    for (int i = 0; i < 8; i++) { a[i] = threadIdx.x * 0.001f + i; p[i] = f2{a[i], a[i] + 1.f}; }
    f2 sp = f2{s, s * 1.0001f};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < REP; r++) {
            const int i = r & 7;
            if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(s));
            if (OP == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p[i]) : "v"(sp));
            if (OP == 2) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(s));
            if (OP == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(sp));
            if (OP == 4) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(s));
            if (OP == 5) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(sp));
            if (OP == 6) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
            if (OP == 7) asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0]" : "+v"(p[i]) : "v"(sp));
        }
    }
    float acc = 0;
    for (int i = 0; i < 8; i++) acc += a[i] + p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
template <int OP> void run(const char* name, float* d, int wgs_per_cu) {
    const int iters = 2000, grid = 256 * wgs_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(256), 0, 0, d, 10, 1.0001f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(256), 0, 0, d, iters, 1.0001f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // wave-instructions per SIMD: each WG = 4 waves, one per SIMD; wgs_per_cu waves per SIMD
    double winstr_per_simd = (double)iters * REP * wgs_per_cu;
    double ns_per = ms * 1e6 / winstr_per_simd;
    printf("%-22s waves/SIMD=%d  %.3f ms  %.3f ns per wave-instr per SIMD (= %.2f cycles @2.4GHz)\n", name, wgs_per_cu, ms, ns_per, ns_per * 2.4);
}
int main() {
    float* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    for (int w : {1, 2, 4}) {
        if (w == 1) { run<0>("v_fma_f32", d, 1); run<1>("v_pk_fma_f32", d, 1); run<2>("v_add_f32", d, 1); run<3>("v_pk_add_f32", d, 1); run<4>("v_mul_f32", d, 1); run<5>("v_pk_mul_f32", d, 1); run<6>("v_mov_b32", d, 1); run<7>("v_pk_add_f32 op_sel", d, 1); }
        if (w == 2) { run<0>("v_fma_f32", d, 2); run<1>("v_pk_fma_f32", d, 2); run<2>("v_add_f32", d, 2); run<3>("v_pk_add_f32", d, 2); run<6>("v_mov_b32", d, 2); }
        if (w == 4) { run<0>("v_fma_f32", d, 4); run<1>("v_pk_fma_f32", d, 4); run<2>("v_add_f32", d, 4); run<3>("v_pk_add_f32", d, 4); run<6>("v_mov_b32", d, 4); }
    }
    return 0;
}
