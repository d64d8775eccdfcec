// acq_kernels.hip -- gfx950 kernels of the GPS L1 C/A acquisition engine.
//
// Replaces the hot loops of c/search_offline.cpp (reference, /root/reference):
//   k_fwd2<bits>                Sample()    :141-161  (unpack, XOR mix, FFT-40000)
//   k_fwd<real>                 SearchInit():101-106  (code replica -> code spectrum)
//   k_corr<MC>                  Correlate() :181-196  (shifted conj-multiply, IFFT-40000,
//                                                      |.|^2 max/argmax/sum over FS/1000 lags)
//   k_peaks                     Correlate() :196-200  (best SNR over the Doppler bins)
//
// Design: one workgroup (256 threads, 4 waves) owns one (block, PRN, Doppler)
// cell.  The 40000-point inverse transform is 8 polyphase 5000-point transforms done in a
// 45 KB LDS buffer (radix 10 x 25 x 20, in place, 16-byte pass-1 stores / pass-3 reads), whose outputs are
// rotated and accumulated in registers; only the FS/1000 lags the reference scans are ever
// formed and nothing but 16 bytes per cell is written back.  The two spectra a cell reads
// are shared by all Doppler bins of a (block, PRN) pair and stay in the XCD's L2: cells of
// one pair are mapped to one XCD.
#include <hip/hip_runtime.h>
#include <type_traits>

#include <cstdlib>

#include "acq_launch.hpp"
#include "acq_phases.hpp"

namespace acq {

// wq[q][m] = W_160^{q m}: the wave-uniform part of the radix-8 rotation.  Constant address space
// so that hipcc reads it with scalar loads and feeds the packed FMAs from SGPR pairs.
__constant__ cf c_wq[NPOLY * WQ_STRIDE];
hipError_t upload_wq(const cf* host) { return hipMemcpyToSymbol(HIP_SYMBOL(c_wq), host, sizeof(cf) * NPOLY * WQ_STRIDE); }

// ---------------------------------------------------------------------------------------
// Forward transform (Sample() :141-161 / SearchInit() :101-106): grid (n_items), one workgroup per item computes the
// eight rows of its polyphase spectrum one after the other and writes each once, coalesced.
// k_fwd:  SRC_REAL: float code replicas (SearchInit); SRC_REALMIX: multi-bit samples.
// k_fwd2: SRC_BITS: the 1-bit capture gps_test reads; SRC_IQ8: an 8-bit IQ capture (rtl-sdr / HackRF) -- mean removal, mixer, sign and
// the bit transpose happen while the block is staged, so the 1-bit stream the reference's MATLAB scripts write to disk
// (proc_rtl_bin_for_gps.m:22-26,43-47) is never materialised.
enum { SRC_REAL = 0, SRC_BITS = 1, SRC_IQ8 = 2, SRC_REALMIX = 3 };  // SRC_REALMIX: multi-bit samples, complex floats with the LO applied

// iq8 staging, step 1: the block's first 5000 bytes of the 1-bit stream, made from the IQ bytes (one aligned 16-byte group
// of 8 samples -> one byte; iq_convert.hpp) into a workgroup-local buffer -- the transform buffer, not yet in use.
// Step 2 is fwd_stage_bits on that buffer.  One byte per loop trip keeps the eight double-precision sincos of a byte
// the only live state (the whole staging unrolled needs 256 VGPRs).
__device__ __forceinline__ void fwd_convert_iq8(int tid, const uint8_t* __restrict__ iq, size_t n0, size_t n_total, const IqConv& c,
                                                uint8_t* stage) {
    const uint4* p = reinterpret_cast<const uint4*>(iq);
#pragma unroll 1
    for (int by = tid; by < USED_BYTES; by += WG) {
        const size_t s0 = n0 + (size_t)by * 8;
        unsigned raw[4] = {0, 0, 0, 0};
        if (s0 + 8 <= n_total) {
            const uint4 v = p[by];
            raw[0] = v.x; raw[1] = v.y; raw[2] = v.z; raw[3] = v.w;
        } else {
            for (size_t sidx = s0; sidx < n_total; ++sidx) {
                const size_t o = 2 * (sidx - n0);
                const unsigned pair = iq[o] | ((unsigned)iq[o + 1] << 8);
                raw[(sidx - s0) >> 1] |= pair << (16 * ((sidx - s0) & 1));
            }
        }
        stage[by] = (uint8_t)iq8_byte(raw, s0, n_total, c);
    }
}

// Float sources (init-time code replicas; multi-bit samples): plain pruned radix-8, forward direction, slot map LayA.
template <int SRC>
__global__ __launch_bounds__(WG) void k_fwd(FwdArgs a) {
    static_assert(SRC == SRC_REAL || SRC == SRC_REALMIX, "k_fwd takes float sources; the 1-bit / IQ captures go through k_fwd2");
    __shared__ cf lds[M_SUB];
    const int tid = threadIdx.x, item = blockIdx.x;
    cf w[2][RA - 1];
    load_tw1(tid, a.t1, w);  // row-independent pass-1 twiddles, once per workgroup
    for (int kappa = 0; kappa < NPOLY; ++kappa) {
        const cf* tn_row = a.tn + (size_t)kappa * M_SUB;
        __syncthreads();  // the previous row's pass-3 reads are done
        if (SRC == SRC_REALMIX) fwd_phase1(tid, kappa, CplxSrc{(const cf*)a.src + (size_t)item * a.src_stride}, tn_row, w, lds);
        else fwd_phase1(tid, kappa, RealSrc{(const float*)a.src + (size_t)item * a.src_stride}, tn_row, w, lds);
        __syncthreads();
        fwd_phase2(tid, a.t2, lds);
        __syncthreads();
        cf y[RC];
        fwd_phase3_load(tid, lds, y);
        fwd_phase3_store(tid, a.conj_out != 0, y, a.out + (size_t)item * a.item_stride + (size_t)kappa * a.row + a.off);
    }
}

// The 1-bit / 8-bit IQ capture (acq_phases.hpp fwd2_*): the block's eight polyphase rows, conjugated; three workgroups per CU
// (46 KB of LDS, <= 168 VGPRs).
template <int SRC>
__global__ __launch_bounds__(WG, 3) void k_fwd2(FwdArgs a) {
    static_assert(SRC == SRC_BITS || SRC == SRC_IQ8, "k_fwd2 is the 1-bit path");
    __shared__ __attribute__((aligned(16))) cf lds[Fwd2Lay::SIZE];  // transform buffer; before the first row: staging of the transposed block
    __shared__ cf t2s[NT2];
    __shared__ cf lutc[256];
    const int tid = threadIdx.x, item = blockIdx.x;
    const int srci = item / a.sub, r = item - srci * a.sub;
    uint8_t* stage = reinterpret_cast<uint8_t*>(lds);            // [5000] converted bytes (iq8 only)
    uint64_t* ib = reinterpret_cast<uint64_t*>(stage + 8192);    // [625]
    uint64_t* qb = reinterpret_cast<uint64_t*>(stage + 16384);   // [625]
    if (SRC == SRC_BITS) fwd_stage_bits(tid, (const uint8_t*)a.src + (size_t)srci * a.src_stride, a.cos_t, a.sin_t, ib, qb);
    if (SRC == SRC_IQ8) {
        fwd_convert_iq8(tid, (const uint8_t*)a.src + (size_t)srci * a.src_stride, a.iq_first + (size_t)srci * (a.src_stride / 2), a.iq_total, a.iq, stage);
        __syncthreads();
        fwd_stage_bits(tid, stage, a.cos_t, a.sin_t, ib, qb);
    }
    for (int i = tid; i < NT2; i += WG) t2s[i] = a.t2[i];
    cf w[2][RA - 1];
    load_tw1<true, Fwd2Lay>(tid, a.t1, w);
    __syncthreads();
    uint32_t packed[RA];
    fwd2_load_bytes(tid, reinterpret_cast<const uint8_t*>(ib), reinterpret_cast<const uint8_t*>(qb), packed);
    const cf* lut_rows = a.lutc + (size_t)r * NPOLY * 256 + tid;  // this thread's entry of each row's look-up table (host-built)
    lutc[tid] = lut_rows[0];
    for (int kappa = 0; kappa < NPOLY; ++kappa) {
        const cf* tn_row = a.tn + ((size_t)r * NPOLY + kappa) * M_SUB;
        const cf next_lut = lut_rows[(kappa + 1 < NPOLY ? kappa + 1 : kappa) * 256];  // requested a row ahead
        __syncthreads();  // table ready; the bytes are in registers (first row) / the previous row's pass-3 reads are done
        fwd2_phase1(tid, packed, lutc, tn_row, w, lds);
        __syncthreads();
        lutc[tid] = next_lut;  // every look-up of this row is done; the next row reads it two barriers on
        fwd2_phase2(tid, t2s, lds);
        __syncthreads();
        fwd2_phase3_store(tid, lds, a.out + (size_t)item * a.item_stride + (size_t)kappa * a.row + a.off);
    }
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
// One workgroup per (task, Doppler grid point).  blockIdx -> cell map keeps the cells of a task on one XCD (block
// b runs on XCD b % 8) and close in time, so both spectra are read from that XCD's L2 (workgroups that each walk many
// cells of different tasks were measured 3..20 % slower: the working set leaves the 4 MB L2, profiles/r02_experiments/h).
// NC = true: non-coherent mode (no reference equivalent): the powers |y[n]|^2 of a.n_acc
// consecutive block spectra (a.acc_step apart) are summed in an LDS array indexed by lag, so that block k's lags can be
// re-aligned by the whole samples the code has crept since block 0 at this cell's Doppler (a.creep samples per block per
// grid point; 0 = plain sum) and by the code phase between block starts that are not whole code periods apart (a.lag_step samples
// per block; 0 = none): power of lag n goes to lag (n - round(k * creep * point) - k * lag_step) mod S.
// W1H: half of the pass-1 twiddles derived instead of held (acq_math.hpp): 18 registers for 18 packed multiplies.
// L: LDS slot map and lane map (acq_math.hpp): LayC (LayB's slots, conflict-free lane assignment) everywhere except the two
// widest non-coherent instances, whose per-lag power array leaves room for the 40 KB map LayA only (33 columns: 77 KB -> still two
// workgroups per CU).
// Wave priority (round 3, profiles/r03_experiments/b_priority_stagger.log): a wave runs the first phase of a sub-transform --
// input loads, product, radix-10 pair, pass-1 stores: the short, latency-bound part that ends in the barrier its three
// partner waves wait at -- one priority level above the long radix-25 / radix-20 phases of the waves of OTHER workgroups it
// shares the SIMD with.  Measured -1.9 % kernel time, repeatably (16.36 vs 16.67 ms per 299 008 cells, same box, two passes);
// levels 1, 2, 3 are equivalent; raising it only while the loads are issued, or also in pass 3, the scan or pass 2, gains
// nothing or less.
#define ACQ_PHASE1_PRIO(level) __builtin_amdgcn_s_setprio(level)
// Peak / sum reduction over the 64 lanes of a wave: inside the four rows of 16 lanes by DPP row shifts (VALU latency), the four
// row results through v_readlane -- instead of six rounds of ds_bpermute (each an LDS-crossbar round trip): -0.5 % kernel time
// (profiles/r03_experiments/g_dpp_reduction.log).
// peak_merge is associative and commutative (larger power, ties to the lower lag), so the order of the merges is free.
template <int CTRL> __device__ __forceinline__ int dpp_mov(int old, int v) { return __builtin_amdgcn_update_dpp(old, v, CTRL, 0xf, 0xf, false); }
__device__ __forceinline__ void wave_reduce_peak(float& mx, int& mi, float& sum) {
#define ACQ_DPP_STEP(CTRL)                                                          \
    {                                                                               \
        const float omx = __int_as_float(dpp_mov<CTRL>(0, __float_as_int(mx)));     \
        const int omi = dpp_mov<CTRL>(0x7fffffff, mi);                              \
        const float os = __int_as_float(dpp_mov<CTRL>(0, __float_as_int(sum)));     \
        peak_merge(mx, mi, omx, omi);                                               \
        sum += os;                                                                  \
    }
    ACQ_DPP_STEP(0x111)  // row_shr:1 -- lane i takes lane i-1 of its row (lanes without a source keep the neutral `old`)
    ACQ_DPP_STEP(0x112)
    ACQ_DPP_STEP(0x114)
    ACQ_DPP_STEP(0x118)
#undef ACQ_DPP_STEP
    // lanes 15, 31, 47, 63 hold their rows' results
    float rmx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mx), 15));
    int rmi = __builtin_amdgcn_readlane(mi, 15);
    float rs = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sum), 15));
#pragma unroll
    for (int r = 1; r < 4; ++r) {
        peak_merge(rmx, rmi, __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mx), 16 * r + 15)), __builtin_amdgcn_readlane(mi, 16 * r + 15));
        rs += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sum), 16 * r + 15));
    }
    mx = rmx;
    mi = rmi;
    sum = rs;
}
// Non-coherent mode, creep re-alignment, more lags than one pass covers (fs > 10 MHz): a lag's re-aligned destination can lie in
// another pass's column window, so the per-lag sums live in device memory instead of this pass's LDS array.  One writer per
// (cell, lag, block); the hardware float add at L2 keeps the read-modify-writes of different waves coherent.
template <int MC>
__device__ __forceinline__ void corr_dump_power(int tid, int rho, int S, int m0, int shift, cf* acc, float* __restrict__ cell_pwr) {
    if (tid >= NBF3) return;
    int sh = shift % S;
    if (sh < 0) sh += S;
#pragma unroll
    for (int m = 0; m < MC; ++m) {
        const int n = NBF3 * (m0 + m) + rho;
        if (n < S) {
            int j = n - sh;
            if (j < 0) j += S;
            unsafeAtomicAdd(cell_pwr + j, acc[m].x * acc[m].x + acc[m].y * acc[m].y);
        }
        acc[m] = mk(0.f, 0.f);
    }
}
template <class L> __device__ __forceinline__ int corr_rho(const CorrArgs& a, int t3) {
    if constexpr (L::REMAP) return (int)a.rho_map[t3];
    else return pass3_rho<L>(t3);
}
// NCREG: non-coherent sums kept in registers (no creep re-alignment asked for: every lag stays with its thread)
// FOLD (round 4; the 22-column coherent instance): the radix-8 rotation W_40000^{q n} of pass 3, n = 250 m + 10 beta + alpha, is split
// into W_4000^{q beta} -- folded into pass 2's output twiddles: a table per sub-transform q -- and W_40000^{q (250 m + alpha)} -- the
// accumulate's factor, read from a [10][columns] LDS table (ten disjoint bank groups; lanes of one alpha broadcast) instead of
// SGPRs.  Pass 3 then multiplies nothing by a per-thread factor: 626 -> 586 VALU instructions per thread and sub-transform.  Both
// tables of the sub-transform in flight are one LDS image refreshed by LDS-DMA (buffer_load ... lds: 1 KB = 64 lanes x 16 bytes per
// wave-instruction, no VGPRs, no ds_write; six of them per sub-transform, spread over the four waves): -1.4 % kernel time on
// configs[1] and [4] (profiles/r04_experiments/a_fold_bq.log; with per-thread copies instead of the DMA it was -0.6 %).
// PERSIST (round 6; every instance that runs three workgroups per CU): the grid is a.persist_wgs workgroups -- as many as are resident
// at once -- and a workgroup walks cells handed out at run time: it draws a ticket from the counter of the XCD it really runs on
// (XCC_ID), and the ticket stands for one Doppler point of one unit of work (a task, or a chunk of a fine grid's task) that the XCD
// took from ONE global counter when it reached it (draw_ticket / task_of_ticket below).  The cells of a unit stay on one XCD and the
// cells in flight on an XCD are consecutive points of one or two units -- the L2 working set of the one-cell-per-workgroup launch (a
// static stride loses it: 24 % slower, profiles/r05_experiments/f_persistent_workgroups.log).  What changes: an XCD takes work at its
// own pace -- under the power cap the XCDs of a package run 3-5 % apart and the fixed deal block g -> XCD g % 8
// moves at the slowest one's, through an in-order dispatcher that leaves slots of the others empty meanwhile -- and a CU never waits
// for a workgroup to be dispatched.  -1.7 ... -4.4 % kernel time by box on configs[1] (profiles/r06_experiments/a_persistent_stealing.log).
// The ticket is drawn before the peak scan and travels to the other threads through the reduction's LDS slots and barrier.
template <int MC, int WPS, int NB, bool NC, bool W1H = false, class L = LayC, bool NCREG = false, bool FOLD = false, bool PERSIST = false>
__global__ __launch_bounds__(WG, WPS) void k_corr(CorrArgs a) {
    static_assert(!PERSIST || !NC || NCREG, "persistent workgroups: the coherent instances and the one with its non-coherent sums in registers");
    __shared__ __attribute__((aligned(16))) cf lds[L::SIZE];  // transform buffer
    // pass 2's 500 twiddles; with FOLD followed by the accumulate factors [alpha][column], in whole 1 KB chunks (the DMA's unit)
    constexpr int TQS = TqStride<MC>::value;
    constexpr int NTAB = NT2 + RA * TQS, NCHUNK = (NTAB + 127) / 128;
    __shared__ __attribute__((aligned(16))) cf tabs[FOLD ? NCHUNK * 128 : NT2];
    cf* const t2s = tabs;
    const cf* const tqs = tabs + NT2;
    __shared__ float red[4 * (WG / 64)];
    __shared__ float pws[NC && !NCREG ? MC * NBF3 : 1];  // non-coherent power per lag of this pass
    const int tid = threadIdx.x;
    // PERSIST: what a cell needs only at its start and end (task list, grid, output, hand-out state) is re-read from the kernel-argument
    // segment where it is used -- scalar loads through a pointer the compiler cannot see through -- instead of living in SGPRs across
    // the sub-transforms, whose radix-25 constants fill the scalar file (it would park them in VGPR lanes: v_readlane in the hot loop)
    typedef const __attribute__((address_space(4))) CorrArgs* KArgs;
    auto K = [&]() __attribute__((always_inline)) {
        KArgs p = (KArgs)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(p));
        return p;
    };
#define KA(field) (PERSIST ? K()->field : a.field)
    // PERSIST: the XCD this workgroup runs on, from the hardware (HW_REG_XCC_ID through __smid(): XCC_ID << 6 | SE_ID << 4 | CU_ID)
    const int my_xcd = PERSIST ? (int)((__smid() >> 6) & 7u) : 0;
    // The next cell of this workgroup's XCD (thread 0 only).  The unit handed to an XCD is a CHUNK of a task: a.persist_chunk
    // consecutive Doppler points (all of them when a task has few -- the reference's 73 bins are one chunk --, ~128 of a fine grid's
    // thousands, so that a few long tasks still spread evenly and no XCD is left with a whole task at the end; an XCD re-reads the
    // task's 650 KB of spectra per chunk: nothing next to the chunk's work).  Ticket t of an XCD is point t % chunk of its
    // (t / chunk)-th unit slot; which unit that is is decided by whoever draws the slot's first ticket -- at once, so that the holders
    // of the slot's other tickets find it published: unit u = the next of ONE global counter = chunk u % units of task u / units.
    auto draw_ticket = [&]() __attribute__((always_inline)) {
        const int t = __hip_atomic_fetch_add(KA(persist_queue) + 16 * my_xcd, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int s, j;
        handout_ticket(t, KA(persist_chunk), s, j);
        if (j == 0 && s < KA(persist_slots)) {
            const int u = __hip_atomic_fetch_add(KA(persist_queue) + 16 * 8, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(KA(persist_tasks) + (size_t)my_xcd * KA(persist_slots) + s, u + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return t;
    };
    // the (task, Doppler point) a ticket stands for; a ticket past the last point of a task's last chunk is void: draw again
    auto task_of_ticket = [&](int ticket, int& di_out) __attribute__((always_inline)) {
        for (;;) {
            int s, j, t;
            handout_ticket(ticket, KA(persist_chunk), s, j);
            if (s >= KA(persist_slots)) return KA(n_tasks);  // (cannot happen: handout_slots)
            int* const slot = KA(persist_tasks) + (size_t)my_xcd * KA(persist_slots) + s;
            int u1;  // (the slot's first ticket was drawn before this one, by a workgroup that published the unit in the same breath)
            while ((u1 = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0) __builtin_amdgcn_s_sleep(1);
            const int kind = handout_cell(u1 - 1, j, KA(persist_units), KA(persist_chunk), KA(ndop), KA(n_tasks), t, di_out);
            if (kind < 0) return KA(n_tasks);
            if (kind > 0) return t;
            ticket = draw_ticket();
        }
    };
    int task, di;
    cf w1[2][RA - 1];
    if constexpr (PERSIST) {
        if (tid == 0) {
            int d0;
            const int t0 = task_of_ticket(draw_ticket(), d0);
            red[3] = __int_as_float(t0);   // (slots 4 w + 3 of the reduction's array are free)
            red[7] = __int_as_float(d0);
        }
        load_tw1<W1H, L>(tid, a.t1, w1);  // once per workgroup
        if constexpr (!FOLD)
            for (int i = tid; i < NT2; i += WG) t2s[i] = a.t2[i];
        __syncthreads();
        task = __builtin_amdgcn_readfirstlane(__float_as_int(red[3]));  // (uniform: kept in SGPRs)
        di = __builtin_amdgcn_readfirstlane(__float_as_int(red[7]));
    } else {
        const int g = blockIdx.x, xcd = g & 7, slot = g >> 3;
        const int grp = slot / KA(ndop);
        di = slot - grp * KA(ndop);
        task = grp * 8 + xcd;
    }
    for (;;) {  // one trip unless PERSIST (every way out of the body returns)
    if (task >= KA(n_tasks)) return;
    const Task tk = KA(tasks)[task];
    // task lists handed over in device memory are not seen by the host: bound them here (uniform branch)
    if (tk.spec < 0 || (long)tk.spec + (long)(KA(n_acc) - 1) * KA(acc_step) >= KA(n_spec) || tk.code < 0 || tk.code >= KA(n_code)) {
        if (tid == 0) {
            Cell c;
            c.max_pwr = 0.f;
            c.max_i = -1;
            c.tot_pwr = 0.f;
            c.snr = 0.f;
            KA(cells)[(size_t)task * KA(ndop) + di] = c;
        }
        if constexpr (PERSIST) {  // (uniform: every thread takes the same next cell through LDS)
            __syncthreads();  // red[3], red[7] of this cell are read
            if (tid == 0) {
                int d0;
                const int t0 = task_of_ticket(draw_ticket(), d0);
                red[3] = __int_as_float(t0);   // (slots 4 w + 3 of the reduction's array are free)
                red[7] = __int_as_float(d0);
            }
            __syncthreads();
            task = __builtin_amdgcn_readfirstlane(__float_as_int(red[3]));  // (uniform: kept in SGPRs)
            di = __builtin_amdgcn_readfirstlane(__float_as_int(red[7]));
            continue;
        } else return;
    }
    int dop, rsub;
    grid_point(di + KA(dop_first), KA(sub), KA(dstride), dop, rsub);
    const cf* dpp = KA(dpp) + ((size_t)tk.spec * KA(sub) + rsub) * NPOLY * M_SUB;
    const cf* cpp = KA(cpp) + (size_t)tk.code * NPOLY * a.crow;

    // q-independent twiddles: pass 2's table into LDS (16-byte copies), pass 1's into registers
    if constexpr (!PERSIST) {
        if constexpr (!FOLD)
            for (int i = tid; i < NT2; i += WG) t2s[i] = a.t2[i];
        load_tw1<W1H, L>(tid, a.t1, w1);
    }

    cf acc[MC];
#pragma unroll
    for (int m = 0; m < MC; ++m) acc[m] = mk(0.f, 0.f);

    if (NC && !NCREG && tid < NBF3) {
#pragma unroll
        for (int m = 0; m < MC; ++m) pws[NBF3 * m + tid] = 0.f;  // first read-modify-write is >= 3 barriers away
    }
    float pw[NCREG ? MC : 1];
    if (NCREG) {
#pragma unroll
        for (int m = 0; m < MC; ++m) pw[m] = 0.f;
    }
    const int t3 = tid < NBF3 ? tid : 0;
    const int rho = corr_rho<L>(a, t3);  // the radix-20 butterfly (output residue) this thread owns
    const int n_acc = NC ? KA(n_acc) : 1;
    // FOLD: which 16 bytes of the global table fold[q] (acq_tables.hpp TablesFold: [t2q (500)][tq (10 x 160)] per q) each lane copies
    // into its slot of the LDS image, for the one or two chunks this wave is responsible for (lanes past the image copy entry 0
    // into the image's padding)
    int fold_wave = 0, fold_voff[2] = {0, 0};
    if constexpr (FOLD) {
        fold_wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#pragma unroll
        for (int sdma = 0; sdma < 2; ++sdma) {
            const int e = 128 * (fold_wave + 4 * sdma) + 2 * (tid & 63);
            int src = 0;
            if (e < NT2) src = e;
            else if (e < NTAB) {
                const int e2 = e - NT2, row = e2 / TQS;
                src = NT2 + row * NW160 + a.m0 + (e2 - row * TQS);
            }
            fold_voff[sdma] = src * (int)sizeof(cf);
        }
    }
    for (int k = 0; k < n_acc; ++k) {
        const cf* dk = dpp + (size_t)k * KA(acc_step) * KA(sub) * NPOLY * M_SUB;
        // one sub-transform; `first` (a std::bool_constant): its outputs are the accumulators' first values
        auto subtransform = [&](const int q, auto first) __attribute__((always_inline)) {
            cf b = mk(0.f, 0.f);
            cf wq_early[(!FOLD && MC <= 22) ? MC : 1];
            const cf* wqv = wq_early;
            if constexpr (FOLD) {
                // this sub-transform's tables, global -> LDS by DMA.  Nobody reads the image any more (pass 2 of the previous
                // sub-transform read t2s before its second barrier, pass 3 read tqs before its third); the barrier that ends phase 1 --
                // which waits for this wave's vmcnt first -- orders the new image before pass 2's reads.
#if defined(__HIP_DEVICE_COMPILE__)
                const __amdgpu_buffer_rsrc_t fold_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.fold, 0, (int)((NPOLY * FOLD_Q + 8) * sizeof(cf)), 0x00020000);
#pragma unroll
                for (int sdma = 0; sdma < 2; ++sdma)
                    if (fold_wave + 4 * sdma < NCHUNK)
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(fold_rsrc, (__attribute__((address_space(3))) void*)(tabs + 128 * (fold_wave + 4 * sdma)), 16,
                                                                 fold_voff[sdma], q * (int)(FOLD_Q * sizeof(cf)), 0, 0);
#endif
            } else {
                b = a.bq[q * NBF3 + rho];  // per-thread rotation of this sub-transform
                // wave-uniform rotations: scalar loads, SGPR operands.  Up to 22 columns are fetched here, ahead of the sub-transform;
                // the wide instances fetch them in chunks inside pass 3 (corr_phase3) to stay inside the SGPR file
                if (MC <= 22) {
#pragma unroll
                    for (int m = 0; m < MC; ++m) wq_early[m] = c_wq[q * WQ_STRIDE + a.m0 + m];
                } else {
                    wqv = c_wq + q * WQ_STRIDE + a.m0;
                }
            }
            ACQ_PHASE1_PRIO(1);
            corr_phase1<NB, W1H, L>(tid, q, dop, dk, cpp, a.crow, a.halo, w1, lds);
            ACQ_PHASE1_PRIO(0);
#if defined(__HIP_DEVICE_COMPILE__)
            // FOLD: the table image written by this wave's LDS-DMA must have landed before the barrier lets another wave's pass 2
            // read it.  The barrier itself does not wait for vector-memory operations, and the only thing that made every wave
            // drain its vmcnt here was that the phase-1 loads (issued after the DMA, returned in order) are consumed above -- a
            // property of today's schedule, not of the source.  The wait is spelled out (it is free: the loads are consumed
            // already) and tools/isa_census.py --assert-dma-wait checks that it is in the ISA.
            if constexpr (FOLD) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
            __syncthreads();  // also orders the t2s fill before its first use
            corr_phase2<L>(tid, t2s, lds);
            __syncthreads();
            constexpr bool FIRST = decltype(first)::value;
            if constexpr (FOLD) corr_phase3_fold<MC, L, FIRST>(tid, rho, tqs, lds, acc);
            else corr_phase3<MC, L, FIRST>(tid, rho, b, wqv, lds, acc);
            __syncthreads();
        };
        // q = 0 carries unit factors (W^0): its iteration is peeled so that the radix-20 outputs land in the accumulators' registers
        // (a branch inside the loop costs a register copy per column instead)
        subtransform(0, std::true_type{});
        for (int q = 1; q < NPOLY; ++q) subtransform(q, std::false_type{});
        if (NC && NCREG) corr_accumulate_power_reg<MC>(tid, acc, pw);
        else if (NC) {
            // every lag has one owner per block, so the scatter needs no atomics; successive blocks'
            // updates are separated by the barriers of the next 8 sub-transforms
            // block k's peak sits later by what the code has crept at this cell's Doppler, and by the code phase its start is ahead of block 0's
            const int shift = __float2int_rn((float)k * a.creep * (float)(di + KA(dop_first))) + k * a.lag_step;
            if (a.pdump) corr_dump_power<MC>(tid, rho, KA(nlags), a.m0, shift, acc, a.pdump + ((size_t)task * KA(ndop) + di) * KA(nlags));
            else corr_accumulate_power<MC>(tid, rho, KA(nlags), a.m0, shift, acc, pws);
        }
    }
    if (NC && !NCREG && a.pdump) return;  // this pass's powers are in a.pdump; launch_scan_power makes the cells

    int next_ticket = 0;  // PERSIST: drawn now, needed after the scan
    if constexpr (PERSIST)
        if (tid == 0) next_ticket = draw_ticket();
    float mx, sum;
    int mi;
    if (NC && NCREG) corr_scan_power_reg<MC>(tid, rho, KA(nlags), a.m0, pw, mx, mi, sum);
    else if (NC) {
        __syncthreads();  // the last block's scatter
        corr_scan_power<MC>(tid, rho, KA(nlags), a.m0, pws, mx, mi, sum);
    }
    else corr_scan<MC>(tid, rho, KA(nlags), a.m0, acc, mx, mi, sum);
    // wave reduction (64 lanes), then across the 4 waves through LDS
    wave_reduce_peak(mx, mi, sum);
    const int wave = tid >> 6;
    if ((tid & 63) == 0) {
        red[wave * 4 + 0] = mx;
        red[wave * 4 + 1] = __int_as_float(mi);
        red[wave * 4 + 2] = sum;
        if (PERSIST && tid == 0) {
            int d0;
            const int t0 = task_of_ticket(next_ticket, d0);
            red[3] = __int_as_float(t0);   // (slots 4 w + 3 of the reduction's array are free)
            red[7] = __int_as_float(d0);
        }
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
        const float ave = sum / (float)KA(nlags);  // :195 tot_pwr / i
        c.snr = (sum > 0.f) ? mx / ave : 0.f;    // :196; 0/0 of the reference defined as 0
        KA(cells)[(size_t)task * KA(ndop) + di] = c;
    }
    // PERSIST: on to this workgroup's next cell.  No barrier needed here: the next cell's first LDS writes (pass-1 stores, the table DMA)
    // touch buffers last read before the final sub-transform's third barrier; `red` is rewritten two dozen barriers from now
    if constexpr (!PERSIST) return;
    task = __builtin_amdgcn_readfirstlane(__float_as_int(red[3]));  // (uniform: kept in SGPRs)
    di = __builtin_amdgcn_readfirstlane(__float_as_int(red[7]));
    }
}
#undef KA

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

// The reference's scan (:190-196) over a cell's per-lag power sums in device memory (corr_dump_power): one workgroup per cell,
// thread t scans lags t, t + 256, ... ascending with a strict '>', the partial results merge with "higher power, ties to the
// lower lag" -- the first maximum, like the serial scan.
__global__ __launch_bounds__(256) void k_scan_power(const float* __restrict__ pdump, Cell* cells, int nlags) {
    if (cells[blockIdx.x].max_i < 0) return;  // a task k_corr rejected (cells were zeroed before the passes): keep its marker
    const float* p = pdump + (size_t)blockIdx.x * nlags;
    float mx = 0.f, sum = 0.f;
    int mi = 0;
    for (int n = threadIdx.x; n < nlags; n += 256) {
        const float v = p[n];
        if (v > mx) { mx = v; mi = n; }
        sum += v;
    }
    wave_reduce_peak(mx, mi, sum);
    __shared__ float red[4 * 4];
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        red[wave * 4 + 0] = mx;
        red[wave * 4 + 1] = __int_as_float(mi);
        red[wave * 4 + 2] = sum;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) {
            peak_merge(mx, mi, red[w * 4 + 0], __float_as_int(red[w * 4 + 1]));
            sum += red[w * 4 + 2];
        }
        Cell c;
        c.max_pwr = mx;
        c.max_i = mi;
        c.tot_pwr = sum;
        const float ave = sum / (float)nlags;
        c.snr = (sum > 0.f) ? mx / ave : 0.f;
        cells[blockIdx.x] = c;
    }
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

// ---------------------------------------------------------------------------------------
// launchers (host)
// (the 1-bit / IQ kernels write the conjugated spectrum, the only one their callers want: run_forward checks)
void launch_fwd_bits(const FwdArgs& a, int n_items, hipStream_t s) {
    hipLaunchKernelGGL(k_fwd2<SRC_BITS>, dim3(n_items), dim3(WG), 0, s, a);
}
void launch_fwd_iq8(const FwdArgs& a, int n_items, hipStream_t s) {
    hipLaunchKernelGGL(k_fwd2<SRC_IQ8>, dim3(n_items), dim3(WG), 0, s, a);
}
void launch_fwd_realmix(const FwdArgs& a, int n_items, hipStream_t s) {
    hipLaunchKernelGGL(k_fwd<SRC_REALMIX>, dim3(n_items), dim3(WG), 0, s, a);
}
void launch_fwd_real(const FwdArgs& a, int n_items, hipStream_t s) {
    hipLaunchKernelGGL(k_fwd<SRC_REAL>, dim3(n_items), dim3(WG), 0, s, a);
}
void launch_code_halo(cf* cpp, int n_rows, int crow, int halo, hipStream_t s) {
    hipLaunchKernelGGL(k_code_halo, dim3(n_rows), dim3(WG), 0, s, cpp, crow, halo);
}
void launch_quirk_patch(const QuirkArgs& a, int n_patch, hipStream_t s) {
    hipLaunchKernelGGL(k_quirk_patch, dim3(n_patch), dim3(WG), 0, s, a);
}
int corr_columns(int nlags) {  // accumulator columns of the smallest instance that covers nlags
    const int need = (nlags + NBF3 - 1) / NBF3;
    const int have[] = {12, 22, 28, 33, 40};
    for (int m : have)
        if (need <= m) return m;
    return MC_MAX;  // several passes of 40 columns
}
bool corr_has_persistent_form(int mc) { return mc == 12 || mc == 22 || mc == 28 || mc == 33; }
int launch_corr(const CorrArgs& a, int mc, hipStream_t s) {
    const int groups = (a.n_tasks + 7) / 8;
    const dim3 grid((unsigned)groups * 8u * (unsigned)a.ndop), block(WG);
    // <columns, waves per SIMD the register allocator is held to (k workgroups per CU <=> k waves per SIMD), load batches,
    // non-coherent, W1H>.  LDS (49 KB per workgroup) admits 3 workgroups per CU; 12, 22 and 28 columns fit 168 VGPRs,
    // 28 and 33 columns do with half the pass-1 twiddles derived (W1H); 40 columns then spill 40 bytes per lane and are still
    // faster at 3 per CU (14.4 vs 13.5 M cells/s; without W1H 116 bytes and slower: profiles/r02_experiments/i, r02j).
    // a.persist_wgs > 0 (the engine armed the hand-out state): the three-per-CU instances run as persistent workgroups
    const bool persist = a.persist_wgs > 0 && a.persist_queue && a.persist_tasks && corr_has_persistent_form(mc);
    const dim3 pgrid((unsigned)(persist ? a.persist_wgs : 1));
    switch (mc) {
        case 12:
            // non-coherent: without creep re-alignment the per-lag sums stay in registers and three workgroups fit a CU (BASELINE
            // configs[3]); with it they go through a per-lag LDS array (61 KB: two per CU)
            if (a.n_acc > 1 && a.creep == 0.f && a.lag_step == 0) {
                if (persist) hipLaunchKernelGGL((k_corr<12, 3, 2, true, false, LayC, true, false, true>), pgrid, block, 0, s, a);
                else hipLaunchKernelGGL((k_corr<12, 3, 2, true, false, LayC, true>), grid, block, 0, s, a);
            } else if (a.n_acc > 1) hipLaunchKernelGGL((k_corr<12, 2, 2, true>), grid, block, 0, s, a);
            else if (persist) hipLaunchKernelGGL((k_corr<12, 3, 2, false, false, LayC, false, false, true>), pgrid, block, 0, s, a);
            else hipLaunchKernelGGL((k_corr<12, 3, 2, false>), grid, block, 0, s, a);
            break;
        case 22:
            if (a.n_acc > 1) hipLaunchKernelGGL((k_corr<22, 2, 2, true>), grid, block, 0, s, a);
            else if (persist) hipLaunchKernelGGL((k_corr<22, 3, 2, false, false, LayC, false, true, true>), pgrid, block, 0, s, a);
            else hipLaunchKernelGGL((k_corr<22, 3, 2, false, false, LayC, false, true>), grid, block, 0, s, a);  // FOLD
            break;
        case 28:
            if (a.n_acc > 1) hipLaunchKernelGGL((k_corr<28, 2, 2, true>), grid, block, 0, s, a);
            else if (persist) hipLaunchKernelGGL((k_corr<28, 3, 2, false, true, LayC, false, false, true>), pgrid, block, 0, s, a);
            else hipLaunchKernelGGL((k_corr<28, 3, 2, false, true>), grid, block, 0, s, a);
            break;
        case 33:
            if (a.n_acc > 1) hipLaunchKernelGGL((k_corr<33, 2, 2, true, false, LayA>), grid, block, 0, s, a);
            else if (persist) hipLaunchKernelGGL((k_corr<33, 3, 2, false, true, LayC, false, false, true>), pgrid, block, 0, s, a);
            else hipLaunchKernelGGL((k_corr<33, 3, 2, false, true>), grid, block, 0, s, a);
            break;
        case 40:
            if (a.n_acc > 1) hipLaunchKernelGGL((k_corr<40, 1, 2, true, false, LayA>), grid, block, 0, s, a);  // 84 KB of LDS: one workgroup per CU
            else hipLaunchKernelGGL((k_corr<40, 3, 2, false, true>), grid, block, 0, s, a);  // 28 bytes of spills: still 7 % faster than 2 per CU; no persistent form (it would spill 84)
            break;
        default: return -1;
    }
    return 0;
}
void launch_scan_power(const float* pdump, Cell* cells, size_t n_cells, int nlags, hipStream_t s) {
    hipLaunchKernelGGL(k_scan_power, dim3((unsigned)n_cells), dim3(256), 0, s, pdump, cells, nlags);
}
void launch_merge_cells(const Cell* parts, Cell* cells, size_t n_cells, int n_parts, int nlags, hipStream_t s) {
    hipLaunchKernelGGL(k_merge_cells, dim3((unsigned)((n_cells + 255) / 256)), dim3(256), 0, s, parts, cells, n_cells, n_parts, nlags);
}
void launch_peaks(const Cell* cells, Peak* peaks, int n_tasks, int ndop, int dop_first, hipStream_t s) {
    const int per_wg = WG / 64;
    hipLaunchKernelGGL(k_peaks, dim3((n_tasks + per_wg - 1) / per_wg), dim3(WG), 0, s, cells, peaks, n_tasks, ndop, dop_first);
}

}  // namespace acq
