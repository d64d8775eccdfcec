// acq_kernels.hip -- gfx950 kernels of the GPS L1 C/A acquisition engine.
//
// Replaces the hot loops of c/search_offline.cpp (reference, /root/reference):
//   k_fwd<bits>                 Sample()    :141-161  (unpack, XOR mix, FFT-40000)
//                               SearchInit():101-106  (code replica -> code spectrum)
//   k_corr<MC>                  Correlate() :181-196  (shifted conj-multiply, IFFT-40000,
//                                                      |.|^2 max/argmax/sum over FS/1000 lags)
//   k_peaks                     Correlate() :196-200  (best SNR over the Doppler bins)
//
// Design: one workgroup (256 threads, 4 waves) owns one (block, PRN, Doppler)
// cell.  The 40000-point inverse transform is 8 polyphase 5000-point transforms done in a
// 40 KB LDS buffer (radix 10 x 25 x 20, in place, conflict-free slot map), whose outputs are
// rotated and accumulated in registers; only the FS/1000 lags the reference scans are ever
// formed and nothing but 16 bytes per cell is written back.  The two spectra a cell reads
// are shared by all Doppler bins of a (block, PRN) pair and stay in the XCD's L2: cells of
// one pair are mapped to one XCD.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <type_traits>

#include "acq_launch.hpp"
#include "acq_phases.hpp"

namespace acq {

// wq[q][m] = W_160^{q m}: the wave-uniform part of the radix-8 rotation.  Constant address space
// so that hipcc reads it with scalar loads and feeds the packed FMAs from SGPR pairs.
__constant__ cf c_wq[NPOLY * WQ_STRIDE];
hipError_t upload_wq(const cf* host) { return hipMemcpyToSymbol(HIP_SYMBOL(c_wq), host, sizeof(cf) * NPOLY * WQ_STRIDE); }

// ---------------------------------------------------------------------------------------
// Forward transform (Sample() :141-161 / SearchInit() :101-106): grid (8, n_items), one workgroup per
// (item, kappa) computes row kappa of the item's polyphase spectrum and writes it once, coalesced.
template <bool BITS>
__global__ __launch_bounds__(WG) void k_fwd(FwdArgs a) {
    __shared__ cf lds[M_SUB];
    __shared__ uint64_t ib[BITS ? USED_BYTES / NPOLY : 1], qb[BITS ? USED_BYTES / NPOLY : 1];
    __shared__ cf lut[BITS ? 256 : 1];
    const int tid = threadIdx.x, kappa = blockIdx.x, item = blockIdx.y;
    const int srci = item / a.sub, r = item - srci * a.sub;
    const cf* tn_row = a.tn + ((size_t)r * NPOLY + kappa) * M_SUB;
    if (BITS) {
        fwd_build_lut(tid, a.rot8 + (r * NPOLY + kappa) * NPOLY, lut);
        fwd_stage_bits(tid, (const uint8_t*)a.src + (size_t)srci * a.src_stride, a.cos_t, a.sin_t, ib, qb);
        __syncthreads();
        fwd_phase1(tid, kappa, BitsSrc{reinterpret_cast<const uint8_t*>(ib), reinterpret_cast<const uint8_t*>(qb), lut}, tn_row, a.t1, lds);
    } else {
        fwd_phase1(tid, kappa, RealSrc{(const float*)a.src + (size_t)srci * a.src_stride}, tn_row, a.t1, lds);
    }
    __syncthreads();
    fwd_phase2(tid, a.t2, lds);
    __syncthreads();
    cf y[RC];
    fwd_phase3_load(tid, lds, y);
    __syncthreads();
    fwd_phase3_store(tid, a.conj_out != 0, y, lds);
    __syncthreads();
    cf* dst = a.out + (size_t)item * a.item_stride + (size_t)kappa * a.row + a.off;
    for (int i = tid; i < M_SUB / 2; i += WG) reinterpret_cast<cf2*>(dst)[i] = reinterpret_cast<const cf2*>(lds)[i];
}

// cyclic halo of the code rows: grid (8 * n_codes), any block size
__global__ void k_code_halo(cf* cpp, int crow, int halo) {
    cf* row = cpp + (size_t)blockIdx.x * crow;
    for (int h = threadIdx.x; h < halo; h += blockDim.x) {
        row[h] = row[M_SUB + h];
        row[halo + M_SUB + h] = row[halo + h];
    }
}

// Reference quirk: Sample() writes 40960 samples into the 40000-entry
// fwd_buf; with g++'s BSS order the last 960 land on code[0][0..959].  For a task whose PRN
// index is 0 this builds a private copy of code 0 with those entries replaced by the block's
// mixed samples 40000..40959.  grid (n_patch), block 256.
__global__ __launch_bounds__(WG) void k_quirk_patch(QuirkArgs a) {
    const int p = blockIdx.x;
    const cf* src = a.code0;
    cf* dst = a.patched + (size_t)p * NPOLY * a.crow;
    const uint8_t* bytes = a.bits + (size_t)a.block_of_patch[p] * a.stride;
    for (int i = threadIdx.x; i < NPOLY * a.crow; i += WG) {
        const int q = i / a.crow, col = i - q * a.crow;
        int j = col - a.halo;  // logical index within the polyphase row, cyclic
        if (j < 0) j += M_SUB;
        if (j >= M_SUB) j -= M_SUB;
        const int k = NPOLY * j + q;  // spectral bin of code[0]
        cf v = src[i];
        if (k < TAIL_SAMPLES) {
            const int byte = USED_BYTES + (k >> 3);
            const unsigned b = bytes[byte];
            const unsigned ib = ((b ^ a.cos_mask[byte]) >> (k & 7)) & 1u, qb = ((b ^ a.sin_mask[byte]) >> (k & 7)) & 1u;
            v = mk(ib ? -1.f : 1.f, qb ? -1.f : 1.f);
        }
        dst[i] = v;
    }
}

// ---------------------------------------------------------------------------------------
// One workgroup per (task, Doppler bin).  blockIdx -> cell map keeps the 2*dmax+1 cells of a
// task on one XCD (block b runs on XCD b % 8) so both spectra are read from that XCD's L2.
// NC = true: non-coherent mode (no reference equivalent): the powers |y[n]|^2
// are summed in an LDS array indexed by lag, so that block k's lags can be re-aligned by the whole
// samples the code has crept since block 0 at this cell's Doppler (a.creep samples per block per bin;
// 0 = plain sum): power of lag n goes to lag (n - round(k * creep * dop)) mod S.
// of a.n_acc consecutive block spectra (a.acc_step apart) are summed per lag before the peak scan.
template <int MC, int WPS, int NB, bool NC>
__global__ __launch_bounds__(WG, WPS) void k_corr(CorrArgs a) {
    __shared__ cf lds[M_SUB];  // transform buffer
    __shared__ cf t2s[NT2];    // the 500 pass-2 twiddles
    __shared__ float pws[NC ? MC * NBF3 : 1];  // non-coherent power per lag of this pass
    const int tid = threadIdx.x;
    const int g = blockIdx.x, xcd = g & 7, slot = g >> 3;
    const int grp = slot / a.ndop, di = slot - grp * a.ndop;
    const int task = grp * 8 + xcd;
    if (task >= a.n_tasks) return;
    const Task tk = a.tasks[task];
    // task lists handed over in device memory are not seen by the host: bound them here (uniform branch)
    if (tk.spec < 0 || (long)tk.spec + (long)(a.n_acc - 1) * a.acc_step >= a.n_spec || tk.code < 0 || tk.code >= a.n_code) {
        if (tid == 0) {
            Cell c;
            c.max_pwr = 0.f;
            c.max_i = -1;
            c.tot_pwr = 0.f;
            c.snr = 0.f;
            a.cells[(size_t)task * a.ndop + di] = c;
        }
        return;
    }
    int dop, rsub;
    grid_point(di + a.dop_first, a.sub, a.dstride, dop, rsub);
    const cf* dpp = a.dpp + ((size_t)tk.spec * a.sub + rsub) * NPOLY * M_SUB;
    const cf* cpp = a.cpp + (size_t)tk.code * NPOLY * a.crow;

    // q-independent twiddles: pass 2's table into LDS (16-byte copies), pass 1's into registers
    for (int i = tid; i < NT2; i += WG) t2s[i] = a.t2[i];
    cf w1[2][RA - 1];
    load_tw1(tid, a.t1, w1);

    cf acc[MC];
#pragma unroll
    for (int m = 0; m < MC; ++m) acc[m] = mk(0.f, 0.f);

    if (NC && tid < NBF3) {
#pragma unroll
        for (int m = 0; m < MC; ++m) pws[NBF3 * m + tid] = 0.f;  // first read-modify-write is >= 3 barriers away
    }
    const int t3 = tid < NBF3 ? tid : 0;
    const int n_acc = NC ? a.n_acc : 1;
    for (int k = 0; k < n_acc; ++k) {
        const cf* dk = dpp + (size_t)k * a.acc_step * a.sub * NPOLY * M_SUB;
        for (int q = 0; q < NPOLY; ++q) {
            const cf b = a.bq[q * NBF3 + t3];  // per-thread rotation of this sub-transform
            cf wqv[MC];                        // wave-uniform rotations: scalar loads, SGPR operands
#pragma unroll
            for (int m = 0; m < MC; ++m) wqv[m] = c_wq[q * WQ_STRIDE + a.m0 + m];
            corr_phase1<NB>(tid, q, dop, dk, cpp, a.crow, a.halo, w1, lds);
            __syncthreads();  // also orders the t2s fill before its first use
            corr_phase2(tid, t2s, lds);
            __syncthreads();
            corr_phase3<MC>(tid, b, wqv, lds, acc);
            __syncthreads();
        }
        if (NC) {
            // every lag has one owner per block, so the scatter needs no atomics; successive blocks'
            // updates are separated by the barriers of the next 8 sub-transforms
            corr_accumulate_power<MC>(tid, a.nlags, a.m0, __float2int_rn((float)k * a.creep * (float)(di + a.dop_first)), acc, pws);
        }
    }

    float mx, sum;
    int mi;
    if (NC) {
        __syncthreads();  // the last block's scatter
        corr_scan_power<MC>(tid, a.nlags, a.m0, pws, mx, mi, sum);
    }
    else corr_scan<MC>(tid, a.nlags, a.m0, acc, mx, mi, sum);
    // wave reduction (64 lanes), then across the 4 waves through LDS
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float omx = __shfl_down(mx, off, 64);
        const int omi = __shfl_down(mi, off, 64);
        const float os = __shfl_down(sum, off, 64);
        peak_merge(mx, mi, omx, omi);
        sum += os;
    }
    float* red = reinterpret_cast<float*>(lds);
    const int wave = tid >> 6;
    if ((tid & 63) == 0) {
        red[wave * 4 + 0] = mx;
        red[wave * 4 + 1] = __int_as_float(mi);
        red[wave * 4 + 2] = sum;
    }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < WG / 64; ++w) {
            peak_merge(mx, mi, red[w * 4 + 0], __float_as_int(red[w * 4 + 1]));
            sum += red[w * 4 + 2];
        }
        Cell c;
        c.max_pwr = mx;
        c.max_i = mi;
        c.tot_pwr = sum;
        const float ave = sum / (float)a.nlags;  // :195 tot_pwr / i
        c.snr = (sum > 0.f) ? mx / ave : 0.f;    // :196; 0/0 of the reference defined as 0
        a.cells[(size_t)task * a.ndop + di] = c;
    }
}

// ---------------------------------------------------------------------------------------
// k_corr2: same arithmetic as k_corr, restructured around what the profiles showed to be the limiter --
// not a pipe but the time a wave spends waiting (L2 latency after every third barrier, barrier skew):
//   * PRE rows (of 10) of the next sub-transform's inputs are requested before the barrier that ends
//     pass 3, the rest after it
//   * LAYB: LDS slot map with 16-byte pass-1 stores and pass-3 reads (acq_math.hpp)
//   * one workgroup walks a contiguous chunk of a task's Doppler bins (a.nchunk workgroups per task):
//     the q-independent twiddles are fetched once per chunk and the next cell's first loads are in
//     flight during the peak scan
// ABL != 0: timing-only ablations (WRONG results; GPSACQ_KVAR 30..): 1 no global loads in the loop, 2 no LDS stores in
// pass 1, 3 pass 2 skipped, 4 no barriers, 5 no pass-3 arithmetic
template <int MC, int WPS, int PRE, bool LAYB, bool PROF, bool PIPE2 = true, int ABL = 0>
__global__ __launch_bounds__(WG, WPS) void k_corr2(CorrArgs a) {
    static_assert(PRE >= 0 && PRE <= RA, "prefetch rows");
    using L = typename std::conditional<LAYB, LayB, LayA>::type;
    __shared__ __attribute__((aligned(16))) cf lds[L::SIZE];
    __shared__ __attribute__((aligned(16))) cf t2s[PIPE2 ? NT2U : NT2];  // pass-2 twiddles (order of use / [beta][j''])
    __shared__ float red[4 * (WG / 64)];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int g = blockIdx.x, xcd = g & 7, slot = g >> 3;
    const int grp = slot / a.nchunk, ch = slot - grp * a.nchunk;
    const int task = grp * 8 + xcd;
    if (task >= a.n_tasks) return;
    const int di0 = (int)(((long)ch * a.ndop) / a.nchunk), di1 = (int)(((long)(ch + 1) * a.ndop) / a.nchunk);
    const Task tk = a.tasks[task];
    if (tk.spec < 0 || tk.spec >= a.n_spec || tk.code < 0 || tk.code >= a.n_code) {
        for (int di = di0 + (int)threadIdx.x; di < di1; di += WG) {
            Cell c;
            c.max_pwr = 0.f;
            c.max_i = -1;
            c.tot_pwr = 0.f;
            c.snr = 0.f;
            a.cells[(size_t)task * a.ndop + di] = c;
        }
        return;
    }
    const int tid = threadIdx.x;
    const cf* dpp0 = a.dpp + (size_t)tk.spec * a.sub * NPOLY * M_SUB;  // the block's first spectrum (sub-bin offset 0)
    const cf* cpp = a.cpp + (size_t)tk.code * NPOLY * a.crow;

    if (PIPE2) {
        for (int i = threadIdx.x; i < NT2U; i += WG) t2s[i] = a.t2u[i];
    } else {
        for (int i = threadIdx.x; i < NT2; i += WG) t2s[i] = a.t2[i];
    }
    cf w1[2][RA - 1];
    load_tw1(tid, a.t1, w1);
    const int t3 = tid < NBF3 ? tid : 0;

    cf2 pd[RA], pc[RA];
    int dop, rsub;
    grid_point(di0 + a.dop_first, a.sub, a.dstride, dop, rsub);
    const cf* dpp = dpp0 + (size_t)rsub * NPOLY * M_SUB;
    if (PRE > 0) corr_issue<0, PRE>(tid, 0, dop, dpp, cpp, a.crow, a.halo, pd, pc);
    unsigned long long tprof[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tm = 0;
#define ACQ_STAMP(k)                                          \
    if (PROF) {                                               \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
        tprof[k] += now_ - tm;                                \
        tm = now_;                                            \
    }
    for (int di = di0; di < di1; ++di) {
        // this cell's and the next cell's grid points (the last sub-transform requests the next cell's first rows)
        int ndop_next, rnext;
        grid_point(di + 1 + a.dop_first, a.sub, a.dstride, ndop_next, rnext);
        const cf* dpp_next = dpp0 + (size_t)rnext * NPOLY * M_SUB;
        cf acc[MC];
#pragma unroll
        for (int m = 0; m < MC; ++m) acc[m] = mk(0.f, 0.f);
        if (PROF) tm = __builtin_amdgcn_s_memtime();
        for (int q = 0; q < NPOLY; ++q) {
            const cf b = a.bq[q * NBF3 + t3];
            cf wqv[MC];
#pragma unroll
            for (int m = 0; m < MC; ++m) wqv[m] = c_wq[q * WQ_STRIDE + a.m0 + m];
            {
                cf x0[RA], x1[RA];
                // rows [0, PRE) were requested before the last barrier; their products and the remaining
                // requests share one scheduling region so that hipcc can interleave them
                if (PRE > 0) corr_mul<0, PRE>(pd, pc, x0, x1);
                if (PRE < RA && !(ABL == 1 && (q > 0 || di > di0))) {
                    constexpr int H = PRE > 0 ? RA : RA / 2;  // no prefetch: two batches like k_corr
                    corr_issue<PRE, H>(tid, q, dop, dpp, cpp, a.crow, a.halo, pd, pc);
                    ACQ_SCHED_FENCE();
                    corr_mul<PRE, H>(pd, pc, x0, x1);
                    if (H < RA) {
                        corr_issue<H, RA>(tid, q, dop, dpp, cpp, a.crow, a.halo, pd, pc);
                        ACQ_SCHED_FENCE();
                        corr_mul<H, RA>(pd, pc, x0, x1);
                    }
                }
                ACQ_STAMP(0);  // inputs loaded and multiplied
                if (ABL == 2) {
                    cf sink = mk(0.f, 0.f);
                    for (int i = 0; i < RA; ++i) sink = sink + x0[i] * w1[0][i % 9] + x1[i] * w1[1][i % 9];
                    if (sink.x == 12345.f) lds[tid] = sink;
                } else
                corr_phase1_store<L>(tid, x0, x1, w1, lds);
            }
            ACQ_STAMP(1);  // pass 1 done
            if (ABL != 4) __syncthreads();
            ACQ_STAMP(2);  // barrier 1
            if (ABL == 3) {}
            else if (PIPE2) { if (tid < NBF2) pass2_pipe<+1, L>(tid, t2s, lds); }
            else if (tid < NBF2) pass2_inplace<+1, L>(tid, t2s, lds);
            ACQ_STAMP(3);  // pass 2 done
            if (ABL != 4) __syncthreads();
            ACQ_STAMP(4);  // barrier 2
            // next sub-transform (or the next cell's first one)
            const int nq = (q + 1) & 7, ndp = (q == NPOLY - 1) ? ndop_next : dop;
            const cf* ndpp = (q == NPOLY - 1) ? dpp_next : dpp;
            // (requested unconditionally: after the chunk's last sub-transform the rows of bin di1 are fetched and dropped --
            // a conditional request would keep the old rows alive through passes 2 and 3 as the other arm of the merge)
            if (ABL == 5) { if (tid < NBF3) acc[q] = acc[q] + lds[tid] * b; }
            else corr_phase3<MC, L>(tid, b, wqv, lds, acc);
            if (PRE > 0) {
                ACQ_SCHED_FENCE();
                corr_issue<0, PRE>(tid, nq, ndp, ndpp, cpp, a.crow, a.halo, pd, pc);
                ACQ_SCHED_FENCE();
            }
            ACQ_STAMP(5);  // pass 3 done
            if (ABL != 4) __syncthreads();
            ACQ_STAMP(6);  // barrier 3
        }
        float mx, sum;
        int mi;
        int tid_scan = tid;  // opaque per cell: keeps hipcc from hoisting the 22 lag indices out of the cell loop (and spilling them)
        asm volatile("" : "+v"(tid_scan));
        corr_scan<MC>(tid_scan, a.nlags, a.m0, acc, mx, mi, sum);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float omx = __shfl_down(mx, off, 64);
            const int omi = __shfl_down(mi, off, 64);
            const float os = __shfl_down(sum, off, 64);
            peak_merge(mx, mi, omx, omi);
            sum += os;
        }
        // the previous cell's reader of `red` is at least 24 barriers behind
        if (lane == 0) {
            red[wave * 4 + 0] = mx;
            red[wave * 4 + 1] = __int_as_float(mi);
            red[wave * 4 + 2] = sum;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            mx = red[0];
            mi = __float_as_int(red[1]);
            sum = red[2];
            for (int w = 1; w < WG / 64; ++w) {
                peak_merge(mx, mi, red[w * 4 + 0], __float_as_int(red[w * 4 + 1]));
                sum += red[w * 4 + 2];
            }
            Cell c;
            c.max_pwr = mx;
            c.max_i = mi;
            c.tot_pwr = sum;
            const float ave = sum / (float)a.nlags;  // :195 tot_pwr / i
            c.snr = (sum > 0.f) ? mx / ave : 0.f;    // :196; 0/0 of the reference defined as 0
            a.cells[(size_t)task * a.ndop + di] = c;
        }
        ACQ_STAMP(7);  // scan + reduction
        dop = ndop_next;
        dpp = dpp_next;
    }
#undef ACQ_STAMP
    if (PROF && a.prof && lane == 0 && (wave == 0 || wave == 3)) {
        for (int k = 0; k < 8; ++k) atomicAdd(a.prof + (wave == 0 ? 0 : 8) + k, tprof[k]);
    }
}

// More than 10000 lags (fs > 10 MHz) take several k_corr passes of 40 columns each; this folds the
// partial cells (ascending lag ranges, so strict '>' keeps the first maximum) and sets the SNR.
__global__ void k_merge_cells(const Cell* parts, Cell* cells, size_t n_cells, int n_parts, int nlags) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_cells) return;
    Cell c = parts[i];
    for (int p = 1; p < n_parts; ++p) {
        const Cell o = parts[(size_t)p * n_cells + i];
        if (o.max_pwr > c.max_pwr) { c.max_pwr = o.max_pwr; c.max_i = o.max_i; }
        c.tot_pwr += o.tot_pwr;
    }
    const float ave = c.tot_pwr / (float)nlags;
    c.snr = (c.tot_pwr > 0.f) ? c.max_pwr / ave : 0.f;
    cells[i] = c;
}

// Best SNR over the Doppler bins of each task: the reference scans dop ascending with a strict '>'
// (:196-198), i.e. the largest SNR wins and ties go to the lowest bin.  One wavefront per task: lane l
// scans bins l, l + 64, ... (ascending, strict '>'), then the lanes merge with the same rule.
__global__ __launch_bounds__(WG) void k_peaks(const Cell* cells, Peak* peaks, int n_tasks, int ndop, int dop_first) {
    const int lane = threadIdx.x & 63;
    const int task = blockIdx.x * (WG / 64) + (threadIdx.x >> 6);
    if (task >= n_tasks) return;
    const Cell* c = cells + (size_t)task * ndop;
    float snr = 0.f;
    int best = -1;  // -1: no bin with snr > 0 (the reference would leave its outputs untouched)
    for (int di = lane; di < ndop; di += 64) {
        const float s = c[di].snr;
        if (s > snr) {
            snr = s;
            best = di;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float os = __shfl_down(snr, off, 64);
        const int ob = __shfl_down(best, off, 64);
        if (ob >= 0 && (os > snr || (os == snr && (best < 0 || ob < best)))) {
            snr = os;
            best = ob;
        }
    }
    if (lane == 0) {
        Peak p;
        p.snr = 0.f;
        p.lo_shift = 0;
        p.ca_shift = 0;
        p.max_pwr = 0.f;
        if (best >= 0) {
            p.snr = snr;
            p.lo_shift = best + dop_first;
            p.ca_shift = c[best].max_i;
            p.max_pwr = c[best].max_pwr;
        }
        peaks[task] = p;
    }
}

// Multi-GPU merge key of a peak: integer MAX over
//   key = snr bits << 32 | (0xFFFF - (lo_shift + kmax)) << 16 | ca_shift
// picks the higher SNR and, on equal SNR, the LOWER Doppler grid point -- the reference's strict '>' scan over
// ascending dop (:196-198).  Non-negative IEEE floats order like their bit patterns.
__global__ void k_pack_keys(const Peak* peaks, unsigned long long* keys, int n, int kmax) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Peak p = peaks[i];
    const unsigned long long snr = (unsigned long long)__float_as_uint(p.snr > 0.f ? p.snr : 0.f);
    const unsigned long long lo = (unsigned long long)(0xFFFF - (p.lo_shift + kmax)) & 0xFFFFull;
    keys[i] = (snr << 32) | (lo << 16) | ((unsigned long long)p.ca_shift & 0xFFFFull);
}

// ---------------------------------------------------------------------------------------
// launchers (host)
void launch_fwd_bits(const FwdArgs& a, int n_items, hipStream_t s) {
    hipLaunchKernelGGL(k_fwd<true>, dim3(NPOLY, n_items), dim3(WG), 0, s, a);
}
void launch_fwd_real(const FwdArgs& a, int n_items, hipStream_t s) {
    hipLaunchKernelGGL(k_fwd<false>, dim3(NPOLY, n_items), dim3(WG), 0, s, a);
}
void launch_code_halo(cf* cpp, int n_rows, int crow, int halo, hipStream_t s) {
    hipLaunchKernelGGL(k_code_halo, dim3(n_rows), dim3(WG), 0, s, cpp, crow, halo);
}
void launch_quirk_patch(const QuirkArgs& a, int n_patch, hipStream_t s) {
    hipLaunchKernelGGL(k_quirk_patch, dim3(n_patch), dim3(WG), 0, s, a);
}
int corr_columns(int nlags) {  // accumulator columns of the smallest instance that covers nlags
    const int need = (nlags + NBF3 - 1) / NBF3;
    const int have[] = {12, 22, 33, 40};
    for (int m : have)
        if (need <= m) return m;
    return MC_MAX;  // several passes of 40 columns
}
// Experiment switch (round-2 kernel work): GPSACQ_KVAR picks the 22-column coherent instance.
static int kvar() {
    static int v = -2;
    if (v == -2) {
        const char* e = getenv("GPSACQ_KVAR");
        v = e ? atoi(e) : -1;
    }
    return v;
}
int corr_chunks(int ndop) {  // workgroups per task for k_corr2
    static int cpw = -1;
    if (cpw < 0) {
        const char* e = getenv("GPSACQ_CPW");
        cpw = e ? atoi(e) : 9;
        if (cpw < 1) cpw = 1;
    }
    return (ndop + cpw - 1) / cpw;
}
int launch_corr(const CorrArgs& a0, int mc, hipStream_t s) {
    CorrArgs a = a0;
    const int groups = (a.n_tasks + 7) / 8;
    const dim3 block(WG);
    const int kv = kvar();
#ifndef ONE_ONLY
    if (mc == 22 && a.n_acc == 1 && kv >= 1) {
        a.nchunk = corr_chunks(a.ndop);
        const dim3 grid2((unsigned)groups * 8u * (unsigned)a.nchunk);
#define K2(PRE, LAYB, PROF, PIPE2) hipLaunchKernelGGL((k_corr2<22, 3, PRE, LAYB, PROF, PIPE2>), grid2, block, 0, s, a)
#define K2A(ABL) hipLaunchKernelGGL((k_corr2<22, 3, 0, false, false, true, ABL>), grid2, block, 0, s, a)
        switch (kv) {
            case 1: K2(0, false, false, false); break;  // k_corr's structure, chunked cells
            case 2: K2(0, false, false, true); break;   // + pipelined pass 2
            case 3: K2(5, false, false, true); break;   // + half the rows requested before the barrier
            case 4: K2(0, true, false, true); break;    // layout B
            case 5: K2(5, true, false, true); break;
            case 6: K2(0, true, false, false); break;   // layout B, pass 2 as in k_corr
            case 31: K2A(1); break;  // ablations (wrong results, timing only)
            case 32: K2A(2); break;
            case 33: K2A(3); break;
            case 34: K2A(4); break;
            case 35: K2A(5); break;
            case 20: K2(0, false, true, true); break;   // s_memtime phase profiles
            case 21: K2(0, true, true, true); break;
            default: return -1;
        }
#undef K2
#undef K2A
        return 0;
    }
#endif
    const dim3 grid((unsigned)groups * 8u * (unsigned)a.ndop);
    // waves per SIMD the register allocator is held to (k workgroups per CU <=> k waves per SIMD):
    // <columns, waves per SIMD the allocator is held to, load batches>.  LDS (44 KB per workgroup)
    // admits 3 workgroups per CU; the two small instances fit 168 VGPRs without spilling.
    switch (mc) {
        case 12:
            if (a.n_acc > 1) hipLaunchKernelGGL((k_corr<12, 2, 2, true>), grid, block, 0, s, a);
            else hipLaunchKernelGGL((k_corr<12, 3, 2, false>), grid, block, 0, s, a);
            break;
        case 22:
            if (a.n_acc > 1) hipLaunchKernelGGL((k_corr<22, 2, 2, true>), grid, block, 0, s, a);
            else hipLaunchKernelGGL((k_corr<22, 3, 2, false>), grid, block, 0, s, a);
            break;
        case 33:
            if (a.n_acc > 1) hipLaunchKernelGGL((k_corr<33, 2, 2, true>), grid, block, 0, s, a);
            else if (getenv("GPSACQ_WIDE3")) hipLaunchKernelGGL((k_corr<33, 3, 2, false>), grid, block, 0, s, a);
            else hipLaunchKernelGGL((k_corr<33, 2, 2, false>), grid, block, 0, s, a);
            break;
        case 40:
            if (a.n_acc > 1) hipLaunchKernelGGL((k_corr<40, 1, 2, true>), grid, block, 0, s, a);  // 84 KB of LDS: one workgroup per CU
            else if (getenv("GPSACQ_WIDE3")) hipLaunchKernelGGL((k_corr<40, 3, 2, false>), grid, block, 0, s, a);
            else hipLaunchKernelGGL((k_corr<40, 2, 2, false>), grid, block, 0, s, a);
            break;
        default: return -1;
    }
    return 0;
}
void launch_merge_cells(const Cell* parts, Cell* cells, size_t n_cells, int n_parts, int nlags, hipStream_t s) {
    hipLaunchKernelGGL(k_merge_cells, dim3((unsigned)((n_cells + 255) / 256)), dim3(256), 0, s, parts, cells, n_cells, n_parts, nlags);
}
void launch_pack_keys(const Peak* peaks, unsigned long long* keys, int n, int kmax, hipStream_t s) {
    hipLaunchKernelGGL(k_pack_keys, dim3((n + 255) / 256), dim3(256), 0, s, peaks, keys, n, kmax);
}
void launch_peaks(const Cell* cells, Peak* peaks, int n_tasks, int ndop, int dop_first, hipStream_t s) {
    const int per_wg = WG / 64;
    hipLaunchKernelGGL(k_peaks, dim3((n_tasks + per_wg - 1) / per_wg), dim3(WG), 0, s, cells, peaks, n_tasks, ndop, dop_first);
}

}  // namespace acq
