// pk_fma_stream -- what the fp32 vector pipe of THIS box sustains on a pure v_pk_fma_f32 stream at k_corr's residency:
// 256-thread workgroups, three per CU (44 KB of LDS each pins that) = three waves per SIMD, no memory traffic, for a few
// seconds, so that the clock is the one the power cap allows under load.  bench.py runs it in its untimed part and reports
// the rate as roofline.pk_fma_stream_TF next to the 157.3 TFLOP/s of the datasheet (2.4 GHz x 4 cycles per packed
// instruction): the ceiling a kernel made of nothing but packed FMAs would reach here (measurement aid, not product code).
// The loop is written on hard-coded registers (256 v_pk_fma_f32 on 8 independent accumulator chains + 3 scalar instructions per
// trip), so what is timed is the instruction stream below and nothing a compiler added.  tools/ubench/gen_pk_bank.py measured
// that the rate does not depend on where the operands live (VGPR bank pairs, SGPR operand, two or three distinct sources).
// usage: pk_fma_stream [seconds]   ->  one JSON line on stdout
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>

#define FMA8                                              \
    "v_pk_fma_f32 v[0:1], v[46:47], v[52:53], v[0:1]\n\t"   \
    "v_pk_fma_f32 v[4:5], v[50:51], v[56:57], v[4:5]\n\t"   \
    "v_pk_fma_f32 v[8:9], v[54:55], v[60:61], v[8:9]\n\t"   \
    "v_pk_fma_f32 v[12:13], v[58:59], v[32:33], v[12:13]\n\t" \
    "v_pk_fma_f32 v[16:17], v[62:63], v[36:37], v[16:17]\n\t" \
    "v_pk_fma_f32 v[20:21], v[34:35], v[40:41], v[20:21]\n\t" \
    "v_pk_fma_f32 v[24:25], v[38:39], v[44:45], v[24:25]\n\t" \
    "v_pk_fma_f32 v[28:29], v[42:43], v[48:49], v[28:29]\n\t"
#define MOV8(b) "v_mov_b32 v" #b ", %1\n\t"
#define FMA32 FMA8 FMA8 FMA8 FMA8
#define REP 256  // packed FMAs per loop trip: the taken branch (an instruction-fetch restart, ~15 cycles) is amortised over 256 of them

template <int LDS_FLOATS> __global__ __launch_bounds__(256) void k_stream(float* out, int iters, float seed) {
    __shared__ float pad[LDS_FLOATS];  // 44 KB: three workgroups per CU, like k_corr (36 KB: four)
    float res;
    asm volatile(
        MOV8(0) MOV8(1) MOV8(4) MOV8(5) MOV8(8) MOV8(9) MOV8(12) MOV8(13) MOV8(16) MOV8(17) MOV8(20) MOV8(21) MOV8(24) MOV8(25) MOV8(28) MOV8(29)
        MOV8(32) MOV8(33) MOV8(34) MOV8(35) MOV8(36) MOV8(37) MOV8(38) MOV8(39) MOV8(40) MOV8(41) MOV8(42) MOV8(43) MOV8(44) MOV8(45) MOV8(46) MOV8(47)
        MOV8(48) MOV8(49) MOV8(50) MOV8(51) MOV8(52) MOV8(53) MOV8(54) MOV8(55) MOV8(56) MOV8(57) MOV8(58) MOV8(59) MOV8(60) MOV8(61) MOV8(62) MOV8(63)
        "s_mov_b32 s22, %2\n\t"
        "1:\n\t" FMA32 FMA32 FMA32 FMA32 FMA32 FMA32 FMA32 FMA32
        "s_sub_u32 s22, s22, 1\n\ts_cmp_lg_u32 s22, 0\n\ts_cbranch_scc1 1b\n\t"
        "v_add_f32 %0, v0, v4\n\tv_add_f32 %0, %0, v9\n\tv_add_f32 %0, %0, v29"
        : "=v"(res)
        : "v"(seed), "s"(iters)
        : "v0", "v1", "v4", "v5", "v8", "v9", "v12", "v13", "v16", "v17", "v20", "v21", "v24", "v25", "v28", "v29", "v32", "v33", "v34", "v35", "v36", "v37",
          "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58",
          "v59", "v60", "v61", "v62", "v63", "s22", "scc");
    out[blockIdx.x * blockDim.x + threadIdx.x] = res + pad[threadIdx.x];
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
    const double tf3 = run<11000>(d, cus, 3, seconds * 0.75, &ns3);
    const double tf4 = run<9000>(d, cus, 4, seconds * 0.25, &ns4);
    printf("{\"pk_fma_stream_TF\": %.3f, \"ns_per_wave_instr_per_simd\": %.4f, \"waves_per_simd\": 3, \"pk_fma_stream_TF_4_waves\": %.3f, "
           "\"ns_per_wave_instr_per_simd_4_waves\": %.4f, \"compute_units\": %d, \"seconds\": %.2f, "
           "\"form\": \"256 x v_pk_fma_f32 v[d], v[a], v[b], v[d] on hard-coded registers (8 independent chains) + 3 scalar instructions per trip\"}\n",
           tf3, ns3, tf4, ns4, cus, seconds);
    (void)hipFree(d);
    return 0;
}
