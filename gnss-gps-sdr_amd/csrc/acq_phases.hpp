// acq_phases.hpp -- per-thread bodies of the acquisition kernels, one function per
// barrier-delimited phase.  A workgroup is 256 threads; `tid` is the thread index.
// acq_kernels.hip wraps these in __global__ kernels (phases separated by __syncthreads());
// tests/emul runs the same functions thread by thread on the CPU to check the index math
// without a GPU (test infrastructure only).
//
// Device data layouts (all complex float, interleaved re/im):
//   spectrum "dpp"  [8][5000]        conj(FFT(block))[8 j + q] stored at [q][j]    (Sample, :121-165)
//   code     "cpp"  [8][5000 + 2H]   FFT(code replica)[8 j + q] at [q][H + j], with a cyclic halo of
//                                    H entries on both sides so that the whole-bin Doppler
//                                    shift (:182) is a plain pointer offset
//
// The vector-memory pipe, not the VALU, was the first limiter of the correlator (ablation in
// DESIGN.md): every global access here is therefore a 16-byte access of two neighbouring
// elements, and the twiddles come from registers (t1), LDS (t2) or scalar loads (wq), never
// from per-lane global loads inside the q loop.
#pragma once
#include <stdint.h>

#include "acq_math.hpp"

namespace acq {

constexpr int WG = 256;           // workgroup size of every kernel here
constexpr int BLOCK_BYTES = 5120; // bytes consumed per Sample() call (10 x 512, :129,135-136)
constexpr int USED_BYTES = 5000;  // 40000 samples actually transformed
constexpr int TAIL_SAMPLES = 960; // samples read past fwd_buf

struct Cell {  // one (block, PRN, Doppler bin) result of Correlate's inner loop (:178-196)
    float max_pwr;
    int32_t max_i;
    float tot_pwr;
    float snr;
};
struct Peak {  // Correlate's return value and out-params (:196-200)
    float snr;
    int32_t lo_shift;
    int32_t ca_shift;
    float max_pwr;
};
struct Task {  // one (block spectrum, code spectrum) pair to search over all Doppler bins
    int32_t spec;
    int32_t code;
};

// Thread tid (< 250) owns the pass-1 butterflies jp and jp + 1, jp = pass1_jp<L>(tid) (2 tid but for LayC).
// Their 2 x 9 twiddles W_5000^{jp alpha}; loaded once per cell by the correlator.
// W1H: only butterfly jp's (the neighbour's are derived, acq_math.hpp pass1_store_pair)
template <bool W1H = false, class L = LayA>
ACQ_HD void load_tw1(int tid, const cf* __restrict__ t1, cf (&w)[2][RA - 1]) {
    if (tid >= NBF3) return;
    const int jp = pass1_jp<L>(tid);
#pragma unroll
    for (int al = 1; al < RA; ++al) {
        if (W1H) w[0][al - 1] = t1[al * NBF1 + jp];
        else ld2(t1 + al * NBF1 + jp, w[0][al - 1], w[1][al - 1]);
    }
}

// ---------------------------------------------------------------------------------------
// Correlate(): prod = conj(data) * shifted code (:181-185) fused into pass 1 of IDFT_5000.
// The 20 loads are issued in NB batches; a scheduling barrier after each batch keeps hipcc from
// sinking them next to their first use (it otherwise emits load, s_waitcnt vmcnt(0), use, load ...).
#if defined(__HIP_DEVICE_COMPILE__)
#define ACQ_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define ACQ_SCHED_FENCE() ((void)0)
#endif
template <int NB, bool W1H = false, class L = LayB>
ACQ_HD void corr_phase1(int tid, int q, int dop, const cf* __restrict__ dpp, const cf* __restrict__ cpp,
                        int crow, int halo, const cf (&w)[2][RA - 1], cf* lds) {
    if (tid >= NBF3) return;
    const int jp = pass1_jp<L>(tid);  // this thread's butterflies jp, jp + 1: elements jp + 500 a, two neighbours per 16-byte load
    int qp, c;
    shift_split(q, dop, qp, c);
    cf x0[RA], x1[RA];
    static_assert(RA % NB == 0, "load batches must divide the 10 rows");
    constexpr int PER = RA / NB;
#if defined(__HIP_DEVICE_COMPILE__)
    // Buffer loads: address = descriptor base + SGPR offset (row of this q / shift, scalar adds) +
    // one 32-bit lane offset, so the 20 loads need no 64-bit vector address arithmetic
    // (global_load's 13-bit immediate cannot span the 8000-byte row stride: 32 v_add_co/v_addc per
    // sub-transform otherwise).  Descriptors cover one block spectrum / one PRN's code rows.
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void*)dpp, 0, NPOLY * M_SUB * (int)sizeof(cf), 0x00020000);
    const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc((void*)cpp, 0, NPOLY * crow * (int)sizeof(cf), 0x00020000);
    const int sd = q * M_SUB * (int)sizeof(cf), sc = (qp * crow + halo + c) * (int)sizeof(cf);
    const int lane = jp * (int)sizeof(cf);
#else
    const cf* drow = dpp + q * M_SUB + jp;
    const cf* crw = cpp + (long)qp * crow + halo + c + jp;
#endif
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        cf2 d[PER], cc[PER];
#pragma unroll
        for (int i = 0; i < PER; ++i) {
#if defined(__HIP_DEVICE_COMPILE__)
            constexpr int ROW = NBF1 * (int)sizeof(cf);
            d[i] = __builtin_bit_cast(cf2, __builtin_amdgcn_raw_buffer_load_b128(rd, lane, sd + ROW * (b * PER + i), 0));
            cc[i] = __builtin_bit_cast(cf2, __builtin_amdgcn_raw_buffer_load_b128(rc, lane, sc + ROW * (b * PER + i), 0));
#else
            d[i] = *reinterpret_cast<const cf2*>(drow + NBF1 * (b * PER + i));
            cc[i] = *reinterpret_cast<const cf2_a8*>(crw + NBF1 * (b * PER + i));  // arbitrary shift: 8-byte aligned only
#endif
        }
        ACQ_SCHED_FENCE();
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            x0[b * PER + i] = cmul(d[i].xy, cc[i].xy);
            x1[b * PER + i] = cmul(d[i].zw, cc[i].zw);
        }
    }
    if (L::SJ == 1) {
        pass1_store_pair<+1, W1H>(x0, x1, jp, w[0], w[1], lds);  // slot maps LayB / LayC (acq_math.hpp): one 16-byte store per alpha
    } else {
        static_assert(L::SJ == 1 || !W1H, "derived twiddles are implemented for slot map LayB");
        pass1_store<+1, L>(x0, jp, w[0], lds);
        pass1_store<+1, L>(x1, jp + 1, w[1], lds);
    }
}

template <class L = LayB>
ACQ_HD void corr_phase2(int tid, const cf* t2, cf* lds) {
    if (tid < NBF2) pass2_inplace<+1, L>(tid, t2, lds);
}
// acc[m] accumulates y[n] for n = 250 (m0 + m) + rho over the 8 polyphase components (m0, the first
// column of this pass, is a multiple of 20 so that column m0 + m reads radix-20 output m % 20):
// W_N^{-q n} = conj(b) (per thread, b = bq[q][rho]) * conj(wqv[m]) (wave-uniform, wqv[m] = W_160^{q m}).
// rho: the radix-20 butterfly this thread owns (pass3_rho<L>(tid); the kernels read LayC's table from device memory).
// FIRST (the sub-transform q = 0 of a block, peeled by the kernel): every factor is W^0 = (1, 0) exactly and the accumulators
// start at zero, so acc[m] = y[m % 20] -- the rotation and the 2 MC accumulate FMAs are skipped, the same values result.
template <int MC, class L = LayB, bool FIRST = false>
ACQ_HD void corr_phase3(int tid, int rho, cf b, const cf* wqv, const cf* lds, cf* acc) {
    if (tid >= NBF3) return;
    cf y[RC];
    pass3_load<+1, L>(rho, lds, y);
    if constexpr (FIRST) {
#pragma unroll
        for (int m = 0; m < MC; ++m) acc[m] = y[m % RC];
        return;
    }
#pragma unroll
    for (int n = 0; n < RC; ++n) y[n] = cmulc(y[n], b);
    // The wave-uniform factors come as SGPR pairs (scalar loads from constant memory in the kernel).  More than ~14 columns
    // of them at once (28 SGPRs, next to the radix-25's constants and the buffer descriptors) overflow the 102 SGPRs and
    // hipcc parks the rest in VGPR lanes (90 v_readlane per sub-transform in the 33-column instance): wide instances take
    // them in chunks, each chunk's scalar loads issued before the previous chunk's FMAs.
    constexpr int NCH = (MC + 13) / 14, CH = (MC + NCH - 1) / NCH;
#pragma unroll
    for (int c0 = 0; c0 < MC; c0 += CH) {
        cf w[CH];
#pragma unroll
        for (int i = 0; i < CH; ++i) w[i] = wqv[c0 + i < MC ? c0 + i : 0];
        if (NCH > 1) ACQ_SCHED_FENCE();
#pragma unroll
        for (int i = 0; i < CH; ++i)
            if (c0 + i < MC) acc[c0 + i] = cmacc_u(acc[c0 + i], y[(c0 + i) % RC], w[i]);
    }
}

// The same with the rotation folded (k_corr<..., FOLD = true>, acq_tables.hpp TablesFold): pass 2 has applied W_4000^{q beta}; what is left,
// W_40000^{q (250 m + alpha)}, depends on the lane only through alpha and is read from the workgroup's LDS copy tqs[alpha][m]
// (row stride TQS: chosen per instance so that the ten rows a wave reads at once sit in disjoint banks; lanes of one alpha
// read one address -- a broadcast).  No per-thread multiply: 20 complex multiplies fewer per thread and sub-transform.
template <int MC> struct TqStride {  // even (16-byte reads), and {2 TQS alpha mod 64, alpha < 10} four banks apart
    static constexpr int value = MC <= 12 ? 14 : MC <= 22 ? 22 : MC <= 28 ? 30 : MC <= 33 ? 34 : 42;
};
template <int MC, class L = LayB, bool FIRST = false>
ACQ_HD void corr_phase3_fold(int tid, int rho, const cf* tqs, const cf* lds, cf* acc) {
    if (tid >= NBF3) return;
    cf y[RC];
    pass3_load<+1, L>(rho, lds, y);
    if constexpr (FIRST) {  // q = 0: tqs holds (1, 0) throughout
#pragma unroll
        for (int m = 0; m < MC; ++m) acc[m] = y[m % RC];
        return;
    }
    const cf* wrow = tqs + (rho % RA) * TqStride<MC>::value;
    constexpr int CH = 8;  // factors fetched per chunk: 4 16-byte reads, 16 VGPRs (requesting a chunk ahead: 168 VGPRs, no gain)
#pragma unroll
    for (int c0 = 0; c0 < MC; c0 += CH) {
        cf w[CH];
#pragma unroll
        for (int i = 0; i < CH; i += 2)
            if (c0 + i < MC) ld2(wrow + c0 + i, w[i], w[i + 1]);  // (the row is padded to an even length)
        ACQ_SCHED_FENCE();
#pragma unroll
        for (int i = 0; i < CH; ++i)
            if (c0 + i < MC) acc[c0 + i] = cmacc(acc[c0 + i], y[(c0 + i) % RC], w[i]);
    }
}

// Doppler grid point k (in units of the grid step) -> whole-bin shift `dop` of the code spectrum (:182) and the
// index r of the sub-bin-offset spectrum of the block.  sub > 1: step = bin / sub, k = dop * sub + r with
// r in [0, sub); else step = dstride bins, k = dop / dstride.  sub = dstride = 1 is the reference's grid.
ACQ_HD void grid_point(int k, int sub, int dstride, int& dop, int& r) {
    if (sub > 1) {
        dop = (k >= 0) ? k / sub : -((-k + sub - 1) / sub);  // floor
        r = k - dop * sub;
    } else {
        dop = k * dstride;
        r = 0;
    }
}

// ---- run-time hand-out of cells to persistent workgroups (k_corr<..., PERSIST>): the index arithmetic, shared by the kernel, the
// engine that sizes the hand-out state and the CPU emulation that walks the protocol (tests/emul) ----
// The unit of work an XCD takes from the global counter is a chunk of one task's Doppler points: the whole task up to 146 points (the
// reference's grids: 73 / 49 / 143 bins), else chunks of ~128, so that a few tasks of thousands of points still spread evenly.
struct Handout {
    int units;  // units per task
    int chunk;  // Doppler points per unit (the task's last unit may hold fewer: tickets past its end are void)
};
ACQ_HD Handout handout_plan(int ndop) {
    Handout h;
    h.units = ndop <= 146 ? 1 : (ndop + 127) / 128;
    h.chunk = (ndop + h.units - 1) / h.units;
    return h;
}
// unit slots per XCD: an XCD draws at most every unit, and its tickets run past the last one by at most one per workgroup
ACQ_HD long handout_slots(long n_tasks, int units, int wgs) { return n_tasks * units + wgs + 2; }
// ticket t of an XCD's counter: point j of the XCD's slot-th unit
ACQ_HD void handout_ticket(int ticket, int chunk, int& slot, int& j) {
    slot = ticket / chunk;
    j = ticket - slot * chunk;
}
// point j of global unit u: the cell (task, di).  1: a cell; 0: void (past the task's last point); -1: past the last task
ACQ_HD int handout_cell(int u, int j, int units, int chunk, int ndop, int n_tasks, int& task, int& di) {
    task = u / units;
    if (task >= n_tasks) return -1;
    di = (u - task * units) * chunk + j;
    return di < ndop ? 1 : 0;
}

// Peak scan over the first S lags (:190-194), this thread's share, ascending n.
template <int MC>
ACQ_HD void corr_scan(int tid, int rho, int S, int m0, const cf* acc, float& mx, int& mi, float& sum) {
    mx = 0.f;
    mi = 0;
    sum = 0.f;
    if (tid >= NBF3) return;
    if (NBF3 * (m0 + MC - 1) <= S) {
        // the instance fits the lag count (uniform test; true for every BASELINE rate): only the last column can reach past S.
        // Six vector instructions per column instead of nine: no bound test, and the winning column is tracked, not its lag
        int mb = -1;
#pragma unroll
        for (int m = 0; m < MC; ++m) {
#pragma clang fp contract(off)  // two products and a sum, like the general path below compiles: the same bits either way
            float p = acc[m].x * acc[m].x + acc[m].y * acc[m].y;
            if (m == MC - 1) p = (NBF3 * (m0 + m) + rho < S) ? p : 0.f;
            const bool up = p > mx;
            mx = up ? p : mx;
            mb = up ? m : mb;
            sum += p;
        }
        mi = mb >= 0 ? NBF3 * (m0 + mb) + rho : 0;
        return;
    }
#pragma unroll
    for (int m = 0; m < MC; ++m) {  // branch-free: lags beyond S contribute a power of 0
        const int n = NBF3 * (m0 + m) + rho;
        const float p = (n < S) ? acc[m].x * acc[m].x + acc[m].y * acc[m].y : 0.f;
        const bool up = p > mx;
        mx = up ? p : mx;
        mi = up ? n : mi;
        sum += p;
    }
}
// non-coherent mode: add this block's powers into the per-lag array pws (lag n of this pass at
// pws[n - 250 m0]), moved down by `shift` whole samples modulo the S lags (shift = 0 when the search
// takes several passes, m0 > 0), and clear the accumulators for the next block
template <int MC>
ACQ_HD void corr_accumulate_power(int tid, int rho, int S, int m0, int shift, cf* acc, float* pws) {
    if (tid >= NBF3) return;
    int sh = shift % S;
    if (sh < 0) sh += S;
#pragma unroll
    for (int m = 0; m < MC; ++m) {
        const int n = NBF3 * (m0 + m) + rho;
        if (n < S) {
            int j = n - sh;
            if (j < 0) j += S;
            pws[j - NBF3 * m0] += acc[m].x * acc[m].x + acc[m].y * acc[m].y;
        }
        acc[m] = mk(0.f, 0.f);
    }
}
// non-coherent mode without lag re-alignment: every lag keeps its owner from block to block, so the sums can stay in registers
// (no per-lag LDS array: the 12-column instance then fits three workgroups per CU like the coherent one)
template <int MC>
ACQ_HD void corr_accumulate_power_reg(int tid, cf* acc, float* pw) {
    if (tid >= NBF3) return;
#pragma unroll
    for (int m = 0; m < MC; ++m) {
        pw[m] += acc[m].x * acc[m].x + acc[m].y * acc[m].y;
        acc[m] = mk(0.f, 0.f);
    }
}
template <int MC>
ACQ_HD void corr_scan_power_reg(int tid, int rho, int S, int m0, const float* pw, float& mx, int& mi, float& sum) {
    mx = 0.f;
    mi = 0;
    sum = 0.f;
    if (tid >= NBF3) return;
#pragma unroll
    for (int m = 0; m < MC; ++m) {
        const int n = NBF3 * (m0 + m) + rho;
        const float p = (n < S) ? pw[m] : 0.f;
        const bool up = p > mx;
        mx = up ? p : mx;
        mi = up ? n : mi;
        sum += p;
    }
}
// same scan as corr_scan over the summed powers
template <int MC>
ACQ_HD void corr_scan_power(int tid, int rho, int S, int m0, const float* pws, float& mx, int& mi, float& sum) {
    mx = 0.f;
    mi = 0;
    sum = 0.f;
    if (tid >= NBF3) return;
#pragma unroll
    for (int m = 0; m < MC; ++m) {
        const int n = NBF3 * (m0 + m) + rho;
        const float p = (n < S) ? pws[n - NBF3 * m0] : 0.f;
        const bool up = p > mx;
        mx = up ? p : mx;
        mi = up ? n : mi;
        sum += p;
    }
}
// strict '>' first-wins of the reference == larger power, ties to the lower lag
ACQ_HD void peak_merge(float& mx, int& mi, float omx, int omi) {
    if (omx > mx || (omx == mx && omi < mi)) { mx = omx; mi = omi; }
}

// ---------------------------------------------------------------------------------------
// Forward transform of Sample() (:141-161) and SearchInit() (:101-106), decimation in frequency:
//   X[8 k' + kappa] = DFT_5000( z_kappa )[k'],
//   z_kappa[n'] = W_N^{n' kappa} * sum_{nu<8} x[n' + 5000 nu] W_8^{nu kappa}
// One workgroup per block walks the eight rows kappa (bits staged and pass-1 twiddles loaded once): row kappa of the
// polyphase spectrum layout comes out in natural order and is written once, already conjugated -- no scratch round trip.  The eight samples of a
// term sit in bytes (n' >> 3) + 625 nu at bit n' & 7 (5000 = 8 * 625).
ACQ_HD float pm1(unsigned bit) {  // 0 -> +1.0f, 1 -> -1.0f   (Bipolar(), :68-70)
    union { unsigned u; float f; } v;
    v.u = 0x3f800000u | (bit << 31);
    return v.f;
}
// 8 x 8 bit-matrix transpose of a 64-bit word, LSB first: bit nu of output byte k = bit k of input byte nu
ACQ_HD uint64_t transpose8x8(uint64_t x) {
    uint64_t t;
    t = (x ^ (x >> 7)) & 0x00AA00AA00AA00AAull;  x = x ^ t ^ (t << 7);
    t = (x ^ (x >> 14)) & 0x0000CCCC0000CCCCull; x = x ^ t ^ (t << 14);
    t = (x ^ (x >> 28)) & 0x00000000F0F0F0F0ull; x = x ^ t ^ (t << 28);
    return x;
}

// 1-bit input (Sample(), :143-153).  Once per workgroup the block is bit-transposed so that byte n'
// holds the eight samples n' + 5000 nu (nu = bit number), with the quadrature-LO masks XOR-ed in
// (transposing commutes with XOR, so the masks are transposed once on the host); a 256-entry
// table lutc[b] = conj(sum_nu (+-1 by bit nu of b) W_8^{nu kappa}) (host-built, acq_tables.hpp) turns the
// radix-8 partial sum of a term into two look-ups: lut[I byte] + i lut[Q byte] (fwd2_phase1).
// real code replica, imag = 0 (SearchInit(), :101-102): init-time only, plain pruned radix-8
struct RealSrc {
    const float* x;      // [40000]
    ACQ_HD cf partial(int np, cf c4, cf c2, cf c1) const {
        cf v[NPOLY];
#pragma unroll
        for (int nu = 0; nu < NPOLY; ++nu) v[nu] = mk(x[np + M_SUB * nu], 0.f);
        return dft8_one(v, c4, c2, c1);
    }
};

// multi-bit samples (8-bit IQ capture kept at full amplitude, SURVEY.md section 8f.1 "direct float path"): complex floats with the
// quadrature LO already applied as signs (iq_kernels.hip::k_iq_to_mixed); plain pruned radix-8 like the code replicas
struct CplxSrc {
    const cf* x;  // [40000]
    ACQ_HD cf partial(int np, cf c4, cf c2, cf c1) const {
        cf v[NPOLY];
#pragma unroll
        for (int nu = 0; nu < NPOLY; ++nu) v[nu] = x[np + M_SUB * nu];
        return dft8_one(v, c4, c2, c1);
    }
};

ACQ_HD void fwd_stage_bits(int tid, const uint8_t* __restrict__ bytes, const uint64_t* __restrict__ cos_t,
                           const uint64_t* __restrict__ sin_t, uint64_t* ib, uint64_t* qb) {
    for (int B = tid; B < USED_BYTES / NPOLY; B += WG) {
        uint64_t x = 0;
#pragma unroll
        for (int nu = 0; nu < NPOLY; ++nu) x |= (uint64_t)bytes[B + (USED_BYTES / NPOLY) * nu] << (8 * nu);
        x = transpose8x8(x);
        ib[B] = x ^ cos_t[B];
        qb[B] = x ^ sin_t[B];
    }
}

template <class Src>
// tn_row: the 5000 decimation-in-frequency twiddles of this row, W_N^{n' (kappa + eps)}
// w: the thread's pass-1 twiddles (load_tw1), row-independent: loaded once per workgroup
ACQ_HD void fwd_phase1(int tid, int kappa, const Src& src, const cf* __restrict__ tn_row, const cf (&w)[2][RA - 1], cf* lds) {
    if (tid >= NBF3) return;
    const cf c4 = w8(4 * kappa), c2 = w8(2 * kappa), c1 = w8(kappa);
    const cf* tk = tn_row + 2 * tid;
    cf x0[RA], x1[RA];
#pragma unroll
    for (int a = 0; a < RA; ++a) {
        cf tw0, tw1;
        ld2(tk + NBF1 * a, tw0, tw1);
        x0[a] = cmul(src.partial(2 * tid + NBF1 * a, c4, c2, c1), tw0);
        x1[a] = cmul(src.partial(2 * tid + 1 + NBF1 * a, c4, c2, c1), tw1);
    }
    pass1_store<-1>(x0, 2 * tid, w[0], lds);
    pass1_store<-1>(x1, 2 * tid + 1, w[1], lds);
}
// ---- the 1-bit path, second form (round 4): the transform runs in the BACKWARD direction on conjugated inputs, so that the
// conjugated spectrum Correlate() wants (:183-184) comes out as it is (no negation pass before the stores); the thread's 40
// sample bytes (the same for all eight rows) live in ten registers instead of being re-read from LDS per row -- the staging
// area is the transform buffer itself -- which makes room in LDS for pass 2's twiddle table; the look-ups of a row are
// issued in two batches of twenty before anything waits for one; half of the pass-1 twiddles are derived (W1H); slot map LayB.
// LDS slot map of k_fwd2: LayB (16-byte pass-1 stores; against LayA -4.5 % forward-stage time, against LayC's conflict-free lane
// maps no difference: profiles/r04_experiments/f_kfwd2_slot_map.log).  Pass 3 keeps the identity butterfly -> thread map whatever
// the slot map: that is what makes the 64 lanes of a wave store 64 consecutive bins.
typedef LayB Fwd2Lay;
// packed[a] = bytes (I[n0], I[n0 + 1], Q[n0], Q[n0 + 1]) of the transposed block at n0 = jp + 500 a, jp = pass1_jp<Fwd2Lay>(tid) the
// first butterfly of the pair the thread owns (2 tid unless the lane map is LayC's)
ACQ_HD void fwd2_load_bytes(int tid, const uint8_t* ib, const uint8_t* qb, uint32_t (&packed)[RA]) {
    if (tid >= NBF3) return;
#pragma unroll
    for (int a = 0; a < RA; ++a) {
        const uint32_t i2 = *reinterpret_cast<const uint16_t*>(ib + pass1_jp<Fwd2Lay>(tid) + NBF1 * a);
        const uint32_t q2 = *reinterpret_cast<const uint16_t*>(qb + pass1_jp<Fwd2Lay>(tid) + NBF1 * a);
        packed[a] = i2 | (q2 << 16);
    }
}
// w0: the thread's pass-1 twiddles W_5000^{2 tid alpha} (load_tw1<true>)
ACQ_HD void fwd2_phase1(int tid, const uint32_t (&packed)[RA], const cf* lutc, const cf* __restrict__ tn_row, const cf (&w)[2][RA - 1], cf* lds) {
    if (tid >= NBF3) return;
    const int jp = pass1_jp<Fwd2Lay>(tid);
    const cf* tk = tn_row + jp;
    cf x0[RA], x1[RA];
    constexpr int HALF = RA / 2;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        cf li0[HALF], li1[HALF], lq0[HALF], lq1[HALF], t0[HALF], t1[HALF];
#pragma unroll
        for (int i = 0; i < HALF; ++i) {
            const uint32_t v = packed[h * HALF + i];
            li0[i] = lutc[v & 0xffu];
            li1[i] = lutc[(v >> 8) & 0xffu];
            lq0[i] = lutc[(v >> 16) & 0xffu];
            lq1[i] = lutc[v >> 24];
            ld2(tk + NBF1 * (h * HALF + i), t0[i], t1[i]);
        }
        ACQ_SCHED_FENCE();
#pragma unroll
        for (int i = 0; i < HALF; ++i) {  // conj(z tn) = conj(z) conj(tn), conj(z) = conj(lut[I]) - i conj(lut[Q])
            x0[h * HALF + i] = cmulc(sub_i(li0[i], lq0[i]), t0[i]);
            x1[h * HALF + i] = cmulc(sub_i(li1[i], lq1[i]), t1[i]);
        }
    }
    pass1_store_pair<+1, true>(x0, x1, jp, w[0], w[1], lds);  // one 16-byte store per alpha, the neighbour's twiddles derived (W1H)
}

ACQ_HD void fwd2_phase2(int tid, const cf* t2s, cf* lds) {
    if (tid < NBF2) pass2_inplace<+1, Fwd2Lay>(tid, t2s, lds);
}
// pass 3 and the stores: output n of the 64 lanes of a wave is 64 consecutive bins k' = 250 n + tid of the row, already conjugated
ACQ_HD void fwd2_phase3_store(int tid, const cf* lds, cf* dst) {
    if (tid >= NBF3) return;
    cf y[RC];
    pass3_load<+1, Fwd2Lay>(tid, lds, y);
#pragma unroll
    for (int n = 0; n < RC; ++n) dst[NBF3 * n + tid] = y[n];
}

ACQ_HD void fwd_phase2(int tid, const cf* __restrict__ t2, cf* lds) {
    if (tid < NBF2) pass2_inplace<-1>(tid, t2, lds);
}
// pass 3 into registers.  Thread tid owns the radix-20 butterfly rho = tid (identity map), so that output n of the 64 lanes of
// a wave is 64 consecutive bins k' = 250 n + rho of the row:
ACQ_HD void fwd_phase3_load(int tid, const cf* lds, cf* y) {
    if (tid < NBF3) pass3_load<-1>(tid, lds, y);
}
// ... and leaves straight from the registers, one 512-byte segment per wave and output (round 3; rounds 1-2 staged the row in
// LDS for 16-byte stores, at the price of two more barriers and an LDS round trip per row).
// natural order k' = 250 n + rho; conjugated for block spectra (Correlate multiplies by conj(data), :183-184)
ACQ_HD void fwd_phase3_store(int tid, bool conj_out, const cf* y, cf* dst) {
    if (tid >= NBF3) return;
#pragma unroll
    for (int n = 0; n < RC; ++n) {
        cf v = y[n];
        if (conj_out) v.y = -v.y;
        dst[NBF3 * n + tid] = v;
    }
}

}  // namespace acq
