// acq_launch.hpp -- argument blocks and host launchers of the kernels in acq_kernels.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "acq_phases.hpp"
#include "iq_convert.hpp"

namespace acq {

struct FwdArgs {
    const void* src;     // bits: packed capture bytes; iq8: interleaved 8-bit I,Q bytes (16-byte aligned); real: float replicas;
                         // realmix: complex floats (multi-bit samples with the LO applied)
    size_t src_stride;   // per item: bytes (bits, iq8), floats (real) or complex floats (realmix)
    IqConv iq;           // iq8 source: format, mean, mixer
    size_t iq_first;     // iq8 source: capture sample index of src's first sample (the mixer's n, proc_rtl_bin_for_gps.m:41)
    size_t iq_total;     // iq8 source: samples of the whole capture (samples beyond it read as bit 0, like the converter's tail)
    const uint64_t* cos_t;  // [625] bit-transposed LO masks (bits source only)
    const uint64_t* sin_t;
    const cf* t1;
    const cf* t2;
    const cf* tn;        // [sub][8][5000] W_N^{n' (kappa + r/sub)}
    const cf* lutc;      // [sub][8][256]  conjugated radix-8 sums by transposed byte (k_fwd2; acq_tables.hpp forward_tables)
    int sub;             // spectra per source item (sub-bin Doppler offsets r/sub, r < sub); item i -> (source i / sub, r = i % sub)
    cf* out;             // [n_items][item_stride], polyphase rows
    size_t item_stride;  // complex elements per item in out
    long row;            // elements per polyphase row in out
    int off;             // halo offset inside a row
    int conj_out;
};
struct QuirkArgs {
    const cf* code0;           // [8][crow] pristine code of PRN index 0
    cf* patched;               // [n_patch][8][crow]
    const uint8_t* bits;       // capture
    size_t stride;             // bytes per block
    const int32_t* block_of_patch;
    const uint8_t* cos_mask;
    const uint8_t* sin_mask;
    int crow, halo;
};
struct CorrArgs {
    const cf* dpp;     // [n_spec][8][5000]
    const cf* cpp;     // [n_code][8][crow]
    const Task* tasks; // [n_tasks]
    const cf* t1;
    const cf* t2;
    const cf* bq;      // [8][250] by rho
    const cf* fold;    // k_corr<..., FOLD>: [8][FOLD_Q] per sub-transform q: pass-2 output twiddles with W_4000^{q beta} folded in [25][20], then the
                       // accumulate factors W_40000^{q (250 m + alpha)} [10][160] (acq_tables.hpp TablesFold)
    const unsigned char* rho_map;  // [256] LayC: pass-3 thread -> rho (acq_math.hpp kRhoC)
    Cell* cells;       // [n_tasks][ndop]
    int n_tasks, ndop, dop_first, nlags, crow, halo;  // bins dop_first .. dop_first+ndop-1
    int m0;               // first accumulator column of this pass (multiple of 40; 0 unless fs > 10 MHz)
    int n_acc, acc_step;  // non-coherent mode: spectra tk.spec + k*acc_step, k < n_acc (n_acc = 1: coherent)
    float creep;          // non-coherent mode: code creep in samples per accumulated block per Doppler bin (0 = off)
    int lag_step;         // non-coherent mode: whole samples of code phase between the starts of accumulated blocks, (block_step *
                          // block samples) mod S, when the blocks are not whole code periods apart and re-alignment is asked for (0 = off)
    int n_spec, n_code;   // rows of dpp / cpp: tasks pointing outside get an empty cell (max_i = -1)
    int sub, dstride;     // Doppler grid (acq_phases.hpp grid_point): dop_first/ndop count grid points; spectrum of (block, r) at row block*sub + r
    int* persist_queue;   // k_corr<..., PERSIST>: 8 per-XCD ticket counters + the global task counter, 64 bytes apart (9 x 16 ints, zeroed before the launch); NULL: off
    int* persist_tasks;   // k_corr<..., PERSIST>: [8][persist_slots] unit (task, chunk) of an XCD's s-th unit slot + 1 (0: not decided yet; zeroed before the launch)
    int persist_slots;    // >= units + workgroups (an XCD's tickets run past the last unit by at most one per workgroup)
    int persist_wgs;      // workgroups of the persistent launch (3 per CU); 0: one workgroup per cell
    int persist_chunk;    // Doppler points per hand-out unit (a chunk of one task): ceil(ndop / persist_units)
    int persist_units;    // units per task: ceil(ndop / ~128)
    float* pdump;      // non-coherent mode with creep re-alignment over several column passes (fs > 10 MHz): per-lag power sums in
                       // device memory, [n_tasks * ndop][nlags], zeroed by the caller; the cells then come from launch_scan_power.  Else NULL
};

// sets this thread's gpsacq_last_error() text and returns `code` (gpsacq_engine.cpp)
__attribute__((visibility("hidden"))) int set_last_error(int code, const char* msg);

void launch_fwd_bits(const FwdArgs& a, int n_items, hipStream_t s);
void launch_fwd_iq8(const FwdArgs& a, int n_items, hipStream_t s);
void launch_fwd_real(const FwdArgs& a, int n_items, hipStream_t s);
void launch_fwd_realmix(const FwdArgs& a, int n_items, hipStream_t s);
void launch_code_halo(cf* cpp, int n_rows, int crow, int halo, hipStream_t s);
void launch_quirk_patch(const QuirkArgs& a, int n_patch, hipStream_t s);
int corr_columns(int nlags);
bool corr_has_persistent_form(int mc);  // instances launch_corr can run as persistent workgroups (CorrArgs::persist_*)
hipError_t upload_wq(const cf* host);  // fills the __constant__ copy of wq on the current device
int launch_corr(const CorrArgs& a, int mc, hipStream_t s);
void launch_scan_power(const float* pdump, Cell* cells, size_t n_cells, int nlags, hipStream_t s);
void launch_merge_cells(const Cell* parts, Cell* cells, size_t n_cells, int n_parts, int nlags, hipStream_t s);
void launch_pack_keys(const Peak* peaks, unsigned long long* keys, int n, int kmax, hipStream_t s);
void launch_cycle_stamp(unsigned long long* out, hipStream_t s);
void launch_prn_keys(const Peak* peaks, int n_tasks, int kmax, unsigned long long* best, hipStream_t s);
void launch_prn_best(const unsigned long long* keys, const Peak* peaks, int n_tasks, unsigned long long* best, float* best_pwr, hipStream_t s);
void launch_max_u64(unsigned long long* dst, const unsigned long long* src, int n, hipStream_t s);
void launch_max_f32(float* dst, const float* src, int n, hipStream_t s);
void launch_winner_pwr(const unsigned long long* own, const unsigned long long* merged, const float* own_pwr, float* out, int n, hipStream_t s);
void launch_peak_pwr(const Peak* peaks, float* pwr, int n, hipStream_t s);
void launch_peaks(const Cell* cells, Peak* peaks, int n_tasks, int ndop, int dop_first, hipStream_t s);

}  // namespace acq
