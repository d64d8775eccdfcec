// pk_fma_stream -- what the fp32 vector pipe of THIS box sustains on a pure v_pk_fma_f32 stream at k_corr's residency:
// 256-thread workgroups, three per CU (44 KB of LDS each pins that) = three waves per SIMD, no memory traffic, for a few
// seconds, so that the clock is the one the power cap allows under load.  bench.py runs it in its untimed part and reports
// the rate as roofline.pk_fma_stream_TF next to the 157.3 TFLOP/s of the datasheet (2.4 GHz x 4 cycles per packed
// instruction): the ceiling a kernel made of nothing but packed FMAs would reach here (measurement aid, not product code).
// The loop is written on hard-coded registers (256 v_pk_fma_f32 on 8 independent accumulator chains + 3 scalar instructions per
// trip), so what is timed is the instruction stream below and nothing a compiler added.  A generator of such streams (round 5, profiles/r05_experiments/a_pk_issue_cost.log) measured
// that the rate does not depend on where the operands live (VGPR bank pairs, SGPR operand, two or three distinct sources).
// The operands ROTATE (below), so that the datapath toggles as it does on real data and the clock is the one a transform kernel gets.
// usage: pk_fma_stream [seconds]   ->  one JSON line on stdout
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>

// One trip of 8: every accumulator pair z (a point near the unit circle, different in every lane) is turned by a small angle,
// z += s * (i z): one packed FMA whose two vector operands change with every step -- multiplier and adder inputs toggle like in a
// transform kernel (with constant multiplicands the same stream draws 250 W less and says nothing about the clock under load).
#define ROT(d) "v_pk_fma_f32 v[" #d "], v[" #d "], v[32:33], v[" #d "] op_sel:[1,0,0] op_sel_hi:[0,0,1] neg_lo:[1,0,0]\n\t"
#define FMA8 ROT(0:1) ROT(4:5) ROT(8:9) ROT(12:13) ROT(16:17) ROT(20:21) ROT(24:25) ROT(28:29)
#define FMA32 FMA8 FMA8 FMA8 FMA8
#define REP 256  // packed FMAs per loop trip: the taken branch (an instruction-fetch restart, ~15 cycles) is amortised over 256 of them

template <int LDS_FLOATS> __global__ __launch_bounds__(256) void k_stream(float* out, int iters, float seed) {
    __shared__ float pad[LDS_FLOATS];  // 44 KB: three workgroups per CU, like k_corr (36 KB: four)
    pad[threadIdx.x] = 0.f;             // (only its own entry is read back below)
    // a different point near the unit circle in every lane (80 000 turns of 2^-10 rad per launch move the radius by 4 %)
    const float ang = 0.001f * (float)(threadIdx.x + 1) + seed, zx = __cosf(ang), zy = __sinf(ang);
    float res;
    asm volatile(
        "v_mov_b32 v0, %1\n\tv_mov_b32 v1, %3\n\tv_mov_b32 v4, %3\n\tv_mov_b32 v5, %1\n\tv_mov_b32 v8, %1\n\tv_mov_b32 v9, %1\n\tv_mov_b32 v12, %3\n\tv_mov_b32 v13, %3\n\t"
        "v_mov_b32 v16, %1\n\tv_mov_b32 v17, %3\n\tv_mov_b32 v20, %3\n\tv_mov_b32 v21, %1\n\tv_mov_b32 v24, %1\n\tv_mov_b32 v25, %1\n\tv_mov_b32 v28, %3\n\tv_mov_b32 v29, %3\n\t"
        "v_mov_b32 v32, 0x3a800000\n\tv_mov_b32 v33, 0x3a800000\n\t"  // s = 2^-10 in both halves
        "s_mov_b32 s22, %2\n\t"
        "1:\n\t" FMA32 FMA32 FMA32 FMA32 FMA32 FMA32 FMA32 FMA32
        "s_sub_u32 s22, s22, 1\n\ts_cmp_lg_u32 s22, 0\n\ts_cbranch_scc1 1b\n\t"
        "v_add_f32 %0, v0, v4\n\tv_add_f32 %0, %0, v9\n\tv_add_f32 %0, %0, v29"
        : "=&v"(res)  // early clobber: never the register of an input
        : "v"(zx), "s"(iters), "v"(zy)
        : "v0", "v1", "v4", "v5", "v8", "v9", "v12", "v13", "v16", "v17", "v20", "v21", "v24", "v25", "v28", "v29", "v32", "v33", "s22", "scc");
    out[blockIdx.x * blockDim.x + threadIdx.x] = res + pad[threadIdx.x];
}

// workgroups of k_stream<LDS_FLOATS> one CU holds at once = waves per SIMD (256 threads = one wave on each of the 4 SIMDs): asked of
// the runtime, so the figure printed is the occupancy found on this device, not the one the 160 KB LDS of gfx950 is assumed to give
template <int LDS_FLOATS> static int resident_workgroups() {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_stream<LDS_FLOATS>, 256, 0) != hipSuccess) return -1;
    return n;
}
template <int LDS_FLOATS> static double run(float* d, int cus, int wps, double seconds, double* ns_per_instr) {
    const int iters = 2500, grid = cus * wps;  // ~10 ms per launch
    hipLaunchKernelGGL(k_stream<LDS_FLOATS>, dim3(grid), dim3(256), 0, 0, d, 100, 1e-6f);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    double total_ms = 0;
    long launches = 0;
    const auto t0 = std::chrono::steady_clock::now();
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
        (void)hipEventRecord(e0);
        for (int k = 0; k < 8; ++k) hipLaunchKernelGGL(k_stream<LDS_FLOATS>, dim3(grid), dim3(256), 0, 0, d, iters, 1e-6f);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        total_ms += ms;
        launches += 8;
    }
    const double instr_per_wave = (double)launches * iters * REP;  // packed FMAs issued by every wave
    const double flops = instr_per_wave * 64.0 * 4.0 * (grid * 4.0);  // 64 lanes x 2 FMAs x 2 flops, grid x 4 waves
    *ns_per_instr = total_ms * 1e6 / (instr_per_wave * wps);  // per wave-instruction per SIMD
    return flops / (total_ms * 1e-3) / 1e12;
}

int main(int argc, char** argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 3.0;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) {
        fprintf(stderr, "pk_fma_stream: no HIP device\n");
        return 2;
    }
    const int cus = prop.multiProcessorCount;
    float* d = nullptr;
    if (hipMalloc((void**)&d, (size_t)cus * 4 * 256 * sizeof(float)) != hipSuccess) return 2;
    double ns3 = 0, ns4 = 0;
    const int occ3 = resident_workgroups<11000>(), occ4 = resident_workgroups<9000>();
    if (occ3 != 3 || occ4 != 4) fprintf(stderr, "pk_fma_stream: occupancy %d / %d workgroups per CU where 3 / 4 were intended (LDS per CU: %zu bytes)\n", occ3, occ4, (size_t)prop.maxSharedMemoryPerMultiProcessor);
    const double tf3 = run<11000>(d, cus, occ3 > 0 ? occ3 : 3, seconds * 0.75, &ns3);
    const double tf4 = run<9000>(d, cus, occ4 > 0 ? occ4 : 4, seconds * 0.25, &ns4);
    printf("{\"pk_fma_stream_TF\": %.3f, \"ns_per_wave_instr_per_simd\": %.4f, \"waves_per_simd\": %d, \"waves_per_simd_second_leg\": %d, \"pk_fma_stream_TF_4_waves\": %.3f, "
           "\"ns_per_wave_instr_per_simd_4_waves\": %.4f, \"compute_units\": %d, \"seconds\": %.2f, "
           "\"form\": \"256 x v_pk_fma_f32 z, z, s, z (z += s i z: 8 independent chains of rotating operands on hard-coded registers) + 3 scalar instructions per trip\"}\n",
           tf3, ns3, occ3, occ4, tf4, ns4, cus, seconds);
    (void)hipFree(d);
    return 0;
}
