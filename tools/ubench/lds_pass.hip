// Micro-benchmark: LDS time of each pass's exact access pattern (3 workgroups of 256 threads per CU like k_corr),
// current slot map slot(al, j'', b) = 500 al + 25 j'' + b against the candidate 564 al + 22 b + j''
// (pass 1: one 16-byte store of two neighbouring butterflies' outputs; pass 3: 16-byte reads).
// Output: ns per pass per workgroup with every CU running 3 workgroups.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float cf __attribute__((ext_vector_type(2)));
typedef float cf2 __attribute__((ext_vector_type(4)));
#define SINK(v) asm volatile("" ::"v"(v))
template <int PAT> __global__ __launch_bounds__(256, 3) void k(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) cf lds[5640];
    const int t = threadIdx.x;
    for (int i = t; i < 5640; i += 256) lds[i] = cf{(float)i, 1.f};
    __syncthreads();
    cf v = cf{(float)t, 2.f};
    for (int it = 0; it < iters; ++it) {
        v.x += 1.f;
        if (PAT == 0 && t < 250) {  // pass 1 today: 2 x 10 ds_write_b64
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int jp = 2 * t + k, b = jp / 20, jpp = jp - 20 * b;
                cf* d = lds + 25 * jpp + b;
#pragma unroll
                for (int al = 0; al < 10; ++al) d[500 * al] = v;
            }
        }
        if (PAT == 1 && t < 250) {  // pass 1 candidate: 10 ds_write_b128
            const int jp = 2 * t, b = jp / 20, jpp = jp - 20 * b;
            cf2* d = reinterpret_cast<cf2*>(lds + 22 * b + jpp);
            const cf2 w = cf2{v.x, v.y, v.y, v.x};
#pragma unroll
            for (int al = 0; al < 10; ++al) d[282 * al] = w;
        }
        if (PAT == 2 && t < 200) {  // pass 2 today: 25 contiguous elements read and written back
            const int al = t / 20, jpp = t - 20 * al;
            cf* p = lds + 500 * al + 25 * jpp;
            cf x[25];
#pragma unroll
            for (int b = 0; b < 25; ++b) x[b] = p[b];
#pragma unroll
            for (int b = 0; b < 25; ++b) p[b] = x[b] + v;
        }
        if (PAT == 3 && t < 200) {  // pass 2 candidate: stride 22
            const int al = t / 20, jpp = t - 20 * al;
            cf* p = lds + 564 * al + jpp;
            cf x[25];
#pragma unroll
            for (int b = 0; b < 25; ++b) x[b] = p[22 * b];
#pragma unroll
            for (int b = 0; b < 25; ++b) p[22 * b] = x[b] + v;
        }
        if (PAT == 4 && t < 250) {  // pass 3 today: 20 reads, stride 25
            const int al = t / 25, be = t - 25 * al;
            const cf* p = lds + 500 * al + be;
#pragma unroll
            for (int j = 0; j < 20; ++j) { cf x = p[25 * j]; SINK(x); }
        }
        if (PAT == 5 && t < 250) {  // pass 3 candidate: 10 ds_read_b128
            const int al = t / 25, be = t - 25 * al;
            const cf2* p = reinterpret_cast<const cf2*>(lds + 564 * al + 22 * be);
#pragma unroll
            for (int j = 0; j < 10; ++j) { cf2 x = p[j]; SINK(x); }
        }
        __syncthreads();
    }
    out[blockIdx.x * 256 + t] = v.x + lds[t].x;
}
template <int PAT> void run(const char* name, float* d) {
    const int iters = 4000, grid = 256 * 3;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<PAT>, dim3(grid), dim3(256), 0, 0, d, 10);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<PAT>, dim3(grid), dim3(256), 0, 0, d, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %.3f ms  %.1f ns per pass per workgroup (3 per CU) = %.0f ns of CU time\n", name, ms, ms * 1e6 / iters, ms * 1e6 / iters / 3);
}
int main() {
    float* d;
    (void)hipMalloc(&d, 256 * 3 * 256 * 4);
    run<0>("pass 1 stores, 20 x b64 (500 al + 25 j + b)", d);
    run<1>("pass 1 stores, 10 x b128 (564 al + 22 b + j)", d);
    run<2>("pass 2 25 reads + 25 writes, contiguous", d);
    run<3>("pass 2 25 reads + 25 writes, stride 22", d);
    run<4>("pass 3 20 reads, stride 25", d);
    run<5>("pass 3 10 x b128 reads", d);
    return 0;
}
