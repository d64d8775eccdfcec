// emul_acq.cpp -- TEST INFRASTRUCTURE.  Runs the per-thread phase functions of
// gnss-gps-sdr_amd/csrc/acq_phases.hpp thread by thread on the CPU, with a std::vector
// standing in for the workgroup's LDS, so that the index math of the HIP kernels can be
// checked against the oracle in the GPU-less authoring container (tests/test_emul.py).
// It is NOT part of the product and is never linked into libgpsacq.so.
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../gnss-gps-sdr_amd/csrc/acq_phases.hpp"
#include "../../gnss-gps-sdr_amd/csrc/acq_tables.hpp"

using namespace acq;

static const Tables& tables() {
    static Tables t;
    return t;
}

// "kernel" k_fwd: one workgroup = one (item, kappa) -> row kappa of the polyphase spectrum
template <class Src>
static void emul_fwd_row(const Src& src, int kappa, bool conj_out, cf* row /*[5000]*/, const cf* tn_row = nullptr) {
    const Tables& T = tables();
    std::vector<cf> lds(M_SUB);
    std::vector<cf> regs((size_t)WG * RC);
    if (!tn_row) tn_row = T.tn.data() + (size_t)kappa * M_SUB;
    for (int tid = 0; tid < WG; ++tid) {
        cf w[2][RA - 1];
        load_tw1(tid, T.t1.data(), w);
        fwd_phase1(tid, kappa, src, tn_row, w, lds.data());
    }
    for (int tid = 0; tid < WG; ++tid) fwd_phase2(tid, T.t2.data(), lds.data());
    for (int tid = 0; tid < WG; ++tid) fwd_phase3_load(tid, lds.data(), &regs[(size_t)tid * RC]);
    for (int tid = 0; tid < WG; ++tid) fwd_phase3_store(tid, conj_out, &regs[(size_t)tid * RC], row);
}

template <class Src>
static void emul_forward_pp(const Src& src, bool conj_out, cf* out, long row, int off) {
    for (int kappa = 0; kappa < NPOLY; ++kappa) emul_fwd_row(src, kappa, conj_out, out + kappa * row + off);
}

static void pp_to_natural(const cf* pp, long row, int off, bool conj, float* out) {
    for (int k = 0; k < N_FFT; ++k) {
        cf v = pp[(k & 7) * row + off + (k >> 3)];
        out[2 * k] = v.x;
        out[2 * k + 1] = conj ? -v.y : v.y;
    }
}

extern "C" {

// Sample(): spectrum of one 5120-byte block in natural order (un-conjugated); spectrum r of `sub` sub-bin Doppler
// offsets (r = 0, sub = 1: the reference's Sample()), through the phases of k_fwd2: bytes in registers, host-built conjugated
// look-up table, the transform run backwards on conjugated inputs, derived pass-1 twiddles.
void emul_forward_bits_sub(const uint8_t* bytes, const uint8_t* cosm, const uint8_t* sinm, int sub, int r, float* out) {
    const Tables& T = tables();
    std::vector<uint64_t> ib(625), qb(625), cos_t(625), sin_t(625);
    transpose_masks(cosm, cos_t.data());
    transpose_masks(sinm, sin_t.data());
    for (int tid = 0; tid < WG; ++tid) fwd_stage_bits(tid, bytes, cos_t.data(), sin_t.data(), ib.data(), qb.data());
    std::vector<cf> tn, lutc;
    forward_tables(sub, tn, &lutc);
    std::vector<cf> pp((size_t)NPOLY * M_SUB), lds(Fwd2Lay::SIZE);
    std::vector<uint32_t> packed((size_t)WG * RA);
    std::vector<cf> w1((size_t)WG * 2 * (RA - 1));
    for (int tid = 0; tid < WG; ++tid) {
        fwd2_load_bytes(tid, reinterpret_cast<const uint8_t*>(ib.data()), reinterpret_cast<const uint8_t*>(qb.data()),
                        *reinterpret_cast<uint32_t(*)[RA]>(&packed[(size_t)tid * RA]));
        load_tw1<true, Fwd2Lay>(tid, T.t1.data(), *reinterpret_cast<cf(*)[2][RA - 1]>(&w1[(size_t)tid * 2 * (RA - 1)]));
    }
    for (int kappa = 0; kappa < NPOLY; ++kappa) {
        const cf* lut = lutc.data() + ((size_t)r * NPOLY + kappa) * 256;
        const cf* tn_row = tn.data() + ((size_t)r * NPOLY + kappa) * M_SUB;
        for (int tid = 0; tid < WG; ++tid)
            fwd2_phase1(tid, *reinterpret_cast<uint32_t(*)[RA]>(&packed[(size_t)tid * RA]), lut, tn_row,
                        *reinterpret_cast<cf(*)[2][RA - 1]>(&w1[(size_t)tid * 2 * (RA - 1)]), lds.data());
        for (int tid = 0; tid < WG; ++tid) fwd2_phase2(tid, T.t2.data(), lds.data());
        for (int tid = 0; tid < WG; ++tid) fwd2_phase3_store(tid, lds.data(), pp.data() + (size_t)kappa * M_SUB);
    }
    pp_to_natural(pp.data(), M_SUB, 0, true, out);  // stored conjugated
}
void emul_forward_bits(const uint8_t* bytes, const uint8_t* cosm, const uint8_t* sinm, float* out) {
    emul_forward_bits_sub(bytes, cosm, sinm, 1, 0, out);
}
// SearchInit(): spectrum of a real 40000-sample replica.
void emul_forward_real(const float* x, float* out) {
    RealSrc src{x};
    std::vector<cf> pp((size_t)NPOLY * M_SUB);
    emul_forward_pp(src, false, pp.data(), M_SUB, 0);
    pp_to_natural(pp.data(), M_SUB, 0, false, out);
}

}  // extern "C"

// One cell of Correlate(): data/code spectra given in natural order (data un-conjugated).
// mc = accumulator columns of the kernel instance (12, 22, 33 or 40).
// w1h: the instance derives half of its pass-1 twiddles (33 and 28 columns in the product).
// lay: 1 = LayB (round 2's lane map), 2 = LayC (the product's: conflict-free lane assignment)
template <class L>
static int emul_cell_l(const float* dspec, const float* cspec, int halo, int dop, int S, int mc, int w1h, float* max_pwr,
                       int* max_i, float* tot_pwr, bool fold = false) {
    const Tables& T = tables();
    static const TablesFold TF;
    const int crow = M_SUB + 2 * halo;
    std::vector<cf> dpp((size_t)NPOLY * M_SUB), cpp((size_t)NPOLY * crow);
    for (int k = 0; k < N_FFT; ++k) {
        dpp[(size_t)(k & 7) * M_SUB + (k >> 3)] = mk(dspec[2 * k], -dspec[2 * k + 1]);
        cpp[(size_t)(k & 7) * crow + halo + (k >> 3)] = mk(cspec[2 * k], cspec[2 * k + 1]);
    }
    for (int q = 0; q < NPOLY; ++q)
        for (int h = 0; h < halo; ++h) {
            cpp[(size_t)q * crow + h] = cpp[(size_t)q * crow + M_SUB + h];
            cpp[(size_t)q * crow + halo + M_SUB + h] = cpp[(size_t)q * crow + halo + h];
        }
    std::vector<cf> lds(L::SIZE);
    std::vector<cf> acc((size_t)WG * MC_MAX, mk(0.f, 0.f));
    std::vector<cf> w1((size_t)WG * 2 * (RA - 1));
    for (int tid = 0; tid < WG; ++tid) {
        auto& w = *reinterpret_cast<cf(*)[2][RA - 1]>(&w1[(size_t)tid * 2 * (RA - 1)]);
        if (w1h) load_tw1<true, L>(tid, T.t1.data(), w);
        else load_tw1<false, L>(tid, T.t1.data(), w);
    }
    for (int q = 0; q < NPOLY; ++q) {
        for (int tid = 0; tid < WG; ++tid) {
            auto& w = *reinterpret_cast<cf(*)[2][RA - 1]>(&w1[(size_t)tid * 2 * (RA - 1)]);
            if (w1h) corr_phase1<2, true, L>(tid, q, dop, dpp.data(), cpp.data(), crow, halo, w, lds.data());
            else corr_phase1<2, false, L>(tid, q, dop, dpp.data(), cpp.data(), crow, halo, w, lds.data());
        }
        const cf* t2_of_q = fold ? TF.t2q.data() + (size_t)q * NT2 : T.t2.data();
        for (int tid = 0; tid < WG; ++tid) corr_phase2<L>(tid, t2_of_q, lds.data());
        if (fold) {  // the workgroup's LDS copy of this sub-transform's accumulate factors, as the kernel lays it out
            std::vector<cf> tqs((size_t)RA * 42, mk(0.f, 0.f));
            auto fill = [&](int tqs_stride) {
                for (int al = 0; al < RA; ++al)
                    for (int c = 0; c < tqs_stride; ++c) tqs[(size_t)al * tqs_stride + c] = TF.tq[((size_t)q * RA + al) * NW160 + c];
            };
            for (int tid = 0; tid < WG; ++tid) {
                cf* a = &acc[(size_t)tid * MC_MAX];
                const int rho = pass3_rho<L>(tid < NBF3 ? tid : 0);
                switch (mc) {
                    case 12: if (!tid) fill(TqStride<12>::value); if (q == 0) corr_phase3_fold<12, L, true>(tid, rho, tqs.data(), lds.data(), a); else corr_phase3_fold<12, L>(tid, rho, tqs.data(), lds.data(), a); break;
                    case 22: if (!tid) fill(TqStride<22>::value); if (q == 0) corr_phase3_fold<22, L, true>(tid, rho, tqs.data(), lds.data(), a); else corr_phase3_fold<22, L>(tid, rho, tqs.data(), lds.data(), a); break;
                    case 28: if (!tid) fill(TqStride<28>::value); if (q == 0) corr_phase3_fold<28, L, true>(tid, rho, tqs.data(), lds.data(), a); else corr_phase3_fold<28, L>(tid, rho, tqs.data(), lds.data(), a); break;
                    case 33: if (!tid) fill(TqStride<33>::value); if (q == 0) corr_phase3_fold<33, L, true>(tid, rho, tqs.data(), lds.data(), a); else corr_phase3_fold<33, L>(tid, rho, tqs.data(), lds.data(), a); break;
                    case 40: if (!tid) fill(TqStride<40>::value); if (q == 0) corr_phase3_fold<40, L, true>(tid, rho, tqs.data(), lds.data(), a); else corr_phase3_fold<40, L>(tid, rho, tqs.data(), lds.data(), a); break;
                    default: return -1;
                }
            }
            continue;
        }
        for (int tid = 0; tid < WG; ++tid) {
            cf* a = &acc[(size_t)tid * MC_MAX];
            const int rho = pass3_rho<L>(tid < NBF3 ? tid : 0);
            const cf b = T.bq[(size_t)q * NBF3 + rho];
            const cf* wq = &T.wq[(size_t)q * WQ_STRIDE];
            switch (mc) {
                case 12: if (q == 0) corr_phase3<12, L, true>(tid, rho, b, wq, lds.data(), a); else corr_phase3<12, L>(tid, rho, b, wq, lds.data(), a); break;
                case 22: if (q == 0) corr_phase3<22, L, true>(tid, rho, b, wq, lds.data(), a); else corr_phase3<22, L>(tid, rho, b, wq, lds.data(), a); break;
                case 28: if (q == 0) corr_phase3<28, L, true>(tid, rho, b, wq, lds.data(), a); else corr_phase3<28, L>(tid, rho, b, wq, lds.data(), a); break;
                case 33: if (q == 0) corr_phase3<33, L, true>(tid, rho, b, wq, lds.data(), a); else corr_phase3<33, L>(tid, rho, b, wq, lds.data(), a); break;
                case 40: if (q == 0) corr_phase3<40, L, true>(tid, rho, b, wq, lds.data(), a); else corr_phase3<40, L>(tid, rho, b, wq, lds.data(), a); break;
                default: return -1;
            }
        }
    }
    float mx = 0.f, sum = 0.f;
    int mi = 0;
    for (int tid = 0; tid < WG; ++tid) {
        float tmx, tsum;
        int tmi;
        const cf* a = &acc[(size_t)tid * MC_MAX];
        const int rho = pass3_rho<L>(tid < NBF3 ? tid : 0);
        switch (mc) {
            case 12: corr_scan<12>(tid, rho, S, 0, a, tmx, tmi, tsum); break;
            case 22: corr_scan<22>(tid, rho, S, 0, a, tmx, tmi, tsum); break;
            case 28: corr_scan<28>(tid, rho, S, 0, a, tmx, tmi, tsum); break;
            case 33: corr_scan<33>(tid, rho, S, 0, a, tmx, tmi, tsum); break;
            default: corr_scan<40>(tid, rho, S, 0, a, tmx, tmi, tsum); break;
        }
        peak_merge(mx, mi, tmx, tmi);
        sum += tsum;
    }
    *max_pwr = mx;
    *max_i = mi;
    *tot_pwr = sum;
    return 0;
}
extern "C" {
int emul_cell(const float* dspec, const float* cspec, int halo, int dop, int S, int mc, int w1h, int lay, float* max_pwr,
              int* max_i, float* tot_pwr) {
    if (lay == 1) return emul_cell_l<LayB>(dspec, cspec, halo, dop, S, mc, w1h, max_pwr, max_i, tot_pwr);
    if (lay == 2) return emul_cell_l<LayC>(dspec, cspec, halo, dop, S, mc, w1h, max_pwr, max_i, tot_pwr);
    if (lay == 3) return emul_cell_l<LayC>(dspec, cspec, halo, dop, S, mc, w1h, max_pwr, max_i, tot_pwr, true);  // folded rotation (k_corr<..., FOLD>)
    return -1;
}
}  // extern "C"

extern "C" {
// the product's lane maps, for the LDS conflict model (tools/lds_maps.py): lay 1 = LayB, 2 = LayC
int emul_pass1_jp(int lay, int t) { return lay == 2 ? pass1_jp<LayC>(t) : pass1_jp<LayB>(t); }
int emul_pass2_owner(int lay, int e) {
    int al, jpp;
    if (lay == 2) pass2_owner<LayC>(e, al, jpp);
    else pass2_owner<LayB>(e, al, jpp);
    return al * 100 + jpp;
}
int emul_pass3_rho(int lay, int t3) { return lay == 2 ? pass3_rho<LayC>(t3) : pass3_rho<LayB>(t3); }
// the lane maps of LayC deal every butterfly of a pass to exactly one lane: 0 if so
int emul_lane_maps_are_permutations(void) {
    bool seen[M_SUB];
    for (bool& b : seen) b = false;
    for (int t = 0; t < NBF3; ++t) {
        const int jp = pass1_jp<LayC>(t);
        if (jp < 0 || jp + 1 >= NBF1 || (jp & 1) || seen[jp]) return 1;
        seen[jp] = true;
    }
    for (int i = 0; i < NBF2; ++i) seen[i] = false;
    for (int e = 0; e < NBF2; ++e) {
        int al, jpp;
        pass2_owner<LayC>(e, al, jpp);
        if (al < 0 || al >= RA || jpp < 0 || jpp >= RC || seen[al * RC + jpp]) return 2;
        seen[al * RC + jpp] = true;
    }
    for (int i = 0; i < NBF3; ++i) seen[i] = false;
    for (int t = 0; t < NBF3; ++t) {
        const int rho = pass3_rho<LayC>(t);
        if (rho < 0 || rho >= NBF3 || seen[rho]) return 3;
        seen[rho] = true;
    }
    return 0;
}

// host-side table/code helpers of the product, exposed for bit-exact checks against the oracle
void emul_code_replica(double fs, int sv, float* out) { code_replica(fs, sv, out); }
void emul_lo_masks(double fc, double fs, uint8_t* cosm, uint8_t* sinm) { lo_masks(fc, fs, BLOCK_BYTES, cosm, sinm); }
int emul_search_code(int sv, int g1) { return search_code(sv, g1); }
// the product's C/A generator (acq_tables.hpp CaCode + kTaps: what gpsacq_create feeds the replicas and the device chip table from)
void emul_ca_chips(int sv, unsigned char* chips /*[1023]*/) {
    CaCode ca(kTaps[sv][0], kTaps[sv][1]);
    for (int i = 0; i < 1023; ++i) {
        chips[i] = (unsigned char)ca.chip();
        ca.clock();
    }
}
// The run-time hand-out of k_corr<..., PERSIST> (acq_kernels.hip draw_ticket / task_of_ticket), walked on the CPU with the product's own
// index arithmetic (acq_phases.hpp handout_*): `wgs` workgroups spread over n_xcd XCDs, each a state machine -- draw a ticket from its
// XCD's counter; if the ticket opens a unit slot, take the next unit of the global counter and publish it (a SEPARATE step: another
// workgroup may look at the slot in between and has to wait, like the kernel's spin); resolve the ticket to a cell, a void or the end --
// scheduled one step at a time in a seeded random order (skew > 0: XCD x is picked (1 + skew x) times as often -- XCDs of different
// speeds).  counts[task * ndop + di] is incremented per cell handed out.  stats: [0] cells, [1] void tickets, [2] waits on an
// unpublished slot, [3] largest slot index used + 1, [4] handout_slots bound, [5] units taken by XCD 0, [6] by the last XCD.
// Returns 0, or 1 if a slot index reached the bound, 2 if the walk did not end within its step budget.
int emul_handout_walk(int n_tasks, int ndop, int wgs, int n_xcd, unsigned seed, int skew, unsigned* counts, long* stats) {
    const Handout h = handout_plan(ndop);
    const long slots = handout_slots(n_tasks, h.units, wgs);
    std::vector<int> ticket_ctr(n_xcd, 0);
    std::vector<int> table((size_t)n_xcd * slots, 0);
    std::vector<long> units_of(n_xcd, 0);
    int next_unit = 0;
    struct Wg { int xcd, state, ticket; };  // state 0 draw, 1 publish, 2 resolve, 3 gone
    std::vector<Wg> wg(wgs);
    for (int w = 0; w < wgs; ++w) wg[w] = Wg{w % n_xcd, 0, 0};
    unsigned long long rng = seed * 2654435761ull + 88172645463325252ull;
    auto rnd = [&]() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return (unsigned)(rng >> 11); };
    long cells = 0, voids = 0, waits = 0, max_slot = 0, alive = wgs;
    const long budget = 64L * ((long)n_tasks * ndop + wgs) + 100000;
    for (long step = 0; alive > 0; ++step) {
        if (step > budget) return 2;
        int w = (int)(rnd() % (unsigned)wgs);
        if (skew > 0 && (int)(rnd() % (unsigned)(1 + skew * (n_xcd - 1))) >= 1 + skew * wg[w].xcd) continue;  // slower XCDs are scheduled less often
        Wg& g = wg[w];
        if (g.state == 3) continue;
        int slot, j;
        if (g.state == 0) {
            g.ticket = ticket_ctr[g.xcd]++;
            handout_ticket(g.ticket, h.chunk, slot, j);
            g.state = (j == 0 && slot < slots) ? 1 : 2;
            continue;
        }
        handout_ticket(g.ticket, h.chunk, slot, j);
        if (slot >= slots) return 1;
        if (slot + 1 > max_slot) max_slot = slot + 1;
        if (g.state == 1) {
            table[(size_t)g.xcd * slots + slot] = ++next_unit;  // unit + 1
            units_of[g.xcd]++;
            g.state = 2;
            continue;
        }
        const int u1 = table[(size_t)g.xcd * slots + slot];
        if (u1 == 0) {
            ++waits;
            continue;
        }
        int task, di;
        const int kind = handout_cell(u1 - 1, j, h.units, h.chunk, ndop, n_tasks, task, di);
        if (kind < 0) {
            g.state = 3;
            --alive;
        } else if (kind == 0) {
            ++voids;
            g.state = 0;
        } else {
            counts[(size_t)task * ndop + di]++;
            ++cells;
            g.state = 0;
        }
    }
    stats[0] = cells; stats[1] = voids; stats[2] = waits; stats[3] = max_slot; stats[4] = slots;
    stats[5] = units_of[0]; stats[6] = units_of[n_xcd - 1];
    return 0;
}
int emul_handout_plan(int ndop, int* units, int* chunk) {
    const Handout h = handout_plan(ndop);
    *units = h.units;
    *chunk = h.chunk;
    return 0;
}
int emul_dmax(double fs, double max_fo) { return doppler_half_range(fs, max_fo); }
int emul_nlags(double fs) { return num_lags(fs); }
}
