// Micro-benchmark: what the fp32 VALU of gfx950 really sustains on THIS kernel's instruction mix.
// valu_rate.hip times one opcode with two distinct registers; here (a) packed ops with three distinct VGPR-pair
// sources / VOP3P modifiers / SGPR sources, (b) the product's own radix-25, radix-20 and radix-10 butterflies
// (acq_math.hpp) iterated on registers with no memory traffic, at 1..3 waves per SIMD (44 KB of LDS per workgroup
// pins the residency like k_corr).  Output: cycles per VALU wave-instruction per SIMD (instruction counts are
// taken from the ISA: build with --save-temps and count, or use the per-kernel numbers printed by tools/ubench/run_mix.sh).
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../gnss-gps-sdr_amd/csrc/acq_math.hpp"
using namespace acq;
#define REP 32
template <int OP> __global__ __launch_bounds__(256, 3) void k_ops(float* out, int iters, float s) {
    __shared__ float pad[11000];  // 44 KB: three workgroups per CU
    cf p[16];
    for (int i = 0; i < 16; i++) p[i] = mk(threadIdx.x * 0.001f + i, threadIdx.x * 0.002f - i);
    const cf sp = mk(s, s * 1.0001f);
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < REP; r++) {
            const int i = r & 7, j = 8 + ((r + 3) & 7), l = 8 + ((r + 5) & 7);
            if (OP == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "v"(p[j]), "v"(p[l]));
            if (OP == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "+v"(p[i]) : "v"(p[j]), "v"(p[l]));
            if (OP == 2) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(p[i]) : "v"(p[j]), "v"(p[l]));
            if (OP == 3) asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(p[i]) : "v"(p[j]), "v"(p[l]));
            if (OP == 4) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(p[i]) : "v"(p[j]), "s"(sp));
            if (OP == 5) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(p[i]) : "v"(p[j]), "v"(p[l]));
            if (OP == 6) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(p[i].x) : "v"(p[j].x), "v"(p[l].y));
            if (OP == 7) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p[i]) : "v"(p[j]));
        }
    }
    cf acc = mk(0.f, 0.f);
    for (int i = 0; i < 16; i++) acc = acc + p[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc.x + acc.y + pad[threadIdx.x];
}
// the product's butterflies on registers
template <int R> __global__ __launch_bounds__(256, 3) void k_bfly(float* out, int iters, float s) {
    __shared__ float pad[11000];
    cf x[25], y[25];
    for (int i = 0; i < 25; i++) x[i] = mk(threadIdx.x * 0.001f + i, threadIdx.x * 0.002f - i);
    const cf w = mk(0.6f, 0.8f);
    for (int it = 0; it < iters; it++) {
        if (R == 25) { radix25<+1>(x, y); for (int i = 0; i < 25; i++) x[i] = cmulc(y[i], w); }
        if (R == 20) { radix20<+1>(x, y); for (int i = 0; i < 20; i++) x[i] = cmulc(y[i], w); }
        if (R == 10) { radix10<+1>(x, y); radix10<+1>(x + 10, y + 10); for (int i = 0; i < 20; i++) x[i] = cmulc(y[i], w); }
        asm volatile("" ::: "memory");
    }
    cf acc = mk(0.f, 0.f);
    for (int i = 0; i < 25; i++) acc = acc + x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc.x + acc.y + pad[threadIdx.x];
}
template <class F> static float timeit(F launch) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch(10);
    hipEventRecord(e0);
    launch(2000);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
int main(int argc, char** argv) {
    float* d; hipMalloc(&d, 256 * 3 * 256 * 4 * 4);
    int clk_khz = 0; hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
    printf("device clock attribute %d kHz\n", clk_khz);
    const char* names[8] = {"pk_fma 3 vgpr-pairs", "pk_fma 3 vgpr + op_sel/neg", "pk_add 2 vgpr-pairs", "pk_add op_sel/neg", "pk_fma sgpr operand", "pk_mul op_sel_hi", "v_fma_f32 3 vgpr", "pk_fma 2 distinct regs"};
    for (int wps : {1, 2, 3}) {
        const int grid = 256 * wps;
#define RUNOP(OP) { float ms = timeit([&](int it) { hipLaunchKernelGGL(k_ops<OP>, dim3(grid), dim3(256), 0, 0, d, it, 1.0001f); }); \
        double n = 2000.0 * REP * wps; printf("waves/SIMD %d  %-28s %.3f ms  %.2f ns per wave-instr per SIMD\n", wps, names[OP], ms, ms * 1e6 / n); }
        RUNOP(0) RUNOP(1) RUNOP(2) RUNOP(3) RUNOP(4) RUNOP(5) RUNOP(6) RUNOP(7)
#define RUNB(R, NINSTR) { float ms = timeit([&](int it) { hipLaunchKernelGGL(k_bfly<R>, dim3(grid), dim3(256), 0, 0, d, it, 1.0001f); }); \
        double n = 2000.0 * wps; printf("waves/SIMD %d  radix-%d + %d rotations          %.3f ms  %.1f ns per iteration per SIMD\n", wps, R, R == 25 ? 25 : 20, ms, ms * 1e6 / n); }
        RUNB(25, 0) RUNB(20, 0) RUNB(10, 0)
    }
    return 0;
}
