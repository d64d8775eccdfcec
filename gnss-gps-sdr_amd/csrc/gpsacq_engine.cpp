// gpsacq_engine.cpp -- host side of libgpsacq.so: device state, batching and the C ABI
// declared in include/gpsacq.h.  All arithmetic of the search runs in the HIP kernels of
// acq_kernels.hip; there is no CPU fallback -- without a gfx950 device every entry point that
// needs one fails with GPSACQ_ERR_DEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <future>
#include <memory>
#include <string>
#include <vector>

#include "../../include/gpsacq.h"
#include "acq_launch.hpp"
#include "acq_tables.hpp"
#include "iq_launch.hpp"
#include "gen_launch.hpp"

using namespace acq;

static_assert(sizeof(gpsacq_cell) == sizeof(Cell) && sizeof(gpsacq_peak) == sizeof(Peak) && sizeof(gpsacq_task) == sizeof(Task),
              "ABI structs must match the kernel structs");

static thread_local std::string g_err;
static int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
// for the other translation units of the library (acq_launch.hpp); not exported
__attribute__((visibility("hidden"))) int acq::set_last_error(int code, const char* msg) {
    g_err = msg ? msg : "";
    return code;
}
#define HIPCHK(expr)                                                                                  \
    do {                                                                                              \
        hipError_t e_ = (expr);                                                                       \
        if (e_ != hipSuccess) return fail(e_ == hipErrorOutOfMemory ? GPSACQ_ERR_NOMEM : GPSACQ_ERR_DEVICE, \
                                          "%s: %s", #expr, hipGetErrorString(e_));                    \
    } while (0)

struct gpsacq_engine {
    gpsacq_params p{};
    int dmax = 0, ndop = 0, dop_first = 0, nlags = 0, mc = 0, halo = 0, crow = 0;  // searched bins: dop_first .. +ndop-1
    int n_acc = 1, acc_step = 0;  // non-coherent accumulation (gpsacq_set_noncoherent)
    // Doppler grid (gpsacq_set_doppler_step): step = bin * dstride / sub, points -kmax..+kmax; sub = dstride = 1 is the reference's
    int sub = 1, dstride = 1, kmax = 0;
    cf* d_lutc = nullptr;  // [sub][8][256] look-up tables of k_fwd2
    bool creep_comp = false;      // re-align accumulated blocks by the code creep of each Doppler bin
    bool block_align = false;     // re-align accumulated blocks by the code phase between their starts (any stride)
    int cus = 0;
    char name[64] = {0};
    hipStream_t stream = nullptr;
    // stage events of the last kTimingRing searches (asynchronous callers read a finished search's
    // times while the next one runs)
    static const int kTimingRing = 8;
    hipEvent_t ev[kTimingRing][4] = {};
    int ring_launches[kTimingRing] = {};
    int64_t ring_cells[kTimingRing] = {};
    long searches = 0;  // searches enqueued so far; search k uses ring slot k % kTimingRing
    // constants
    cf *d_t1 = nullptr, *d_t2 = nullptr, *d_bq = nullptr, *d_tn = nullptr;
    cf* d_fold = nullptr;  // the folded-rotation tables of k_corr<..., FOLD> (acq_tables.hpp TablesFold)
    unsigned char* d_rho = nullptr;
    uint8_t *d_cos = nullptr, *d_sin = nullptr;
    uint64_t *d_cos_t = nullptr, *d_sin_t = nullptr;  // bit-transposed masks for k_fwd
    cf* d_code = nullptr;  // [32 + patch_cap][8][crow]
    size_t patch_cap = 0;
    int32_t* d_patch_blocks = nullptr;
    // scratch (grown on demand)
    uint8_t* d_bits = nullptr;
    size_t bits_cap = 0;
    cf* d_dpp = nullptr;
    size_t dpp_cap = 0;  // in blocks
    Task* d_tasks = nullptr;
    Cell* d_cells = nullptr;
    Cell* d_parts = nullptr;  // partial cells of multi-pass searches (more than 10000 lags)
    Peak* d_peaks = nullptr;
    size_t task_cap = 0, cell_cap = 0, peak_cap = 0, parts_cap = 0;
    // 8-bit IQ ingestion scratch
    uint8_t* d_iq = nullptr;
    size_t iq_cap = 0;
    uint8_t* d_iqbits = nullptr;
    size_t iqbits_cap = 0;
    float* d_pdump = nullptr;  // non-coherent + creep re-alignment at fs > 10 MHz: per-lag power sums, [cell][nlags]
    size_t pdump_cap = 0;
    float* d_fsamp = nullptr;  // multi-bit path: the batch's samples as complex floats, LO applied ([block][40000][2])
    size_t fsamp_cap = 0;
    unsigned long long* d_sums = nullptr;
    // capture generator scratch
    GenSat* d_sats = nullptr;
    size_t sats_cap = 0;
    uint8_t* d_gen = nullptr;
    size_t gen_cap = 0;
    // k_corr<..., PERSIST>: the hand-out state of a launch (9 counters 64 bytes apart, then [8][slots] task slots), zeroed before it
    int* d_persist = nullptr;
    size_t persist_cap = 0;
    bool persist = false;
    // cached default schedule (task t = block t, PRN t % 32; with ref_quirks also its patch list: blocks 0, 32, 64, ...)
    size_t sched_tasks = 0;
    bool sched_valid = false;
    // pipelined host-buffer searches (gpsacq_pipe_*): a second stream for the uploads, per-slot pinned staging and device buffers
    hipStream_t copy_stream = nullptr;
    struct PipeSlot {
        uint8_t* h_in = nullptr;   // pinned
        size_t h_cap = 0;
        uint8_t* d_in = nullptr;
        size_t d_cap = 0;
        Peak* d_peaks = nullptr;
        Peak* h_peaks = nullptr;   // pinned
        size_t peak_cap = 0;
        hipEvent_t uploaded = nullptr, done = nullptr;
        bool busy = false;
        size_t n_tasks = 0;
    } pipe[GPSACQ_PIPE_SLOTS];
};

// what a search transforms: the 1-bit stream gps_test reads, or an 8-bit IQ capture converted while it is staged
struct Capture {
    bool iq8 = false;
    const uint8_t* d_src = nullptr;  // device pointer
    size_t stride = 0;               // bytes per block in d_src
    IqConv iq{};
    size_t iq_first = 0, iq_total = ~(size_t)0;
    int multibit = 0;  // 8-bit IQ kept at full amplitude (float samples) instead of its sign: 1 the real-IF value, 2 the complex sample
};

static const size_t kFwdChunk = 32768;  // blocks per forward-transform launch (grid.y bound)
static const size_t kPdumpBytes = (size_t)1 << 30;  // per-lag power sums of the multi-pass re-alignment path, per chunk of tasks

// Scratch buffers grow on demand, stream-ordered (hipFreeAsync / hipMallocAsync on the engine's stream): work already
// enqueued keeps the old buffer until it has run, and no device-wide synchronisation happens in mid-stream.  They grow by at
// least half so that a caller creeping up in batch size does not reallocate every call.
template <class T> static int grow(T*& p, size_t& cap, size_t need, hipStream_t stream, size_t elem_bytes = sizeof(T)) {
    if (need <= cap) return GPSACQ_OK;
    need = std::max(need, cap + cap / 2);
    if (p) HIPCHK(hipFreeAsync(p, stream));
    p = nullptr;
    cap = 0;
    HIPCHK(hipMallocAsync((void**)&p, need * elem_bytes, stream));
    cap = need;
    return GPSACQ_OK;
}

static int ensure_code_slots(gpsacq_engine* e, size_t n_patch) {
    if (n_patch <= e->patch_cap) return GPSACQ_OK;
    const size_t slot = (size_t)NPOLY * e->crow;
    cf* nd = nullptr;
    HIPCHK(hipMalloc((void**)&nd, (GPSACQ_NUM_SATS + n_patch) * slot * sizeof(cf)));
    HIPCHK(hipMemcpyAsync(nd, e->d_code, GPSACQ_NUM_SATS * slot * sizeof(cf), hipMemcpyDeviceToDevice, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    HIPCHK(hipFree(e->d_code));
    e->d_code = nd;
    e->patch_cap = 0;
    e->sched_valid = false;  // the cached default schedule's patch list lived in the buffer freed below
    if (e->d_patch_blocks) HIPCHK(hipFree(e->d_patch_blocks));
    e->d_patch_blocks = nullptr;
    HIPCHK(hipMalloc((void**)&e->d_patch_blocks, n_patch * sizeof(int32_t)));
    e->patch_cap = n_patch;
    return GPSACQ_OK;
}

// forward transforms of n items into out (polyphase layout)
// (sub spectra per source item when bits / iq8: item i of the grid -> source i / sub, sub-bin offset i % sub;
//  sub_override > 0: that many instead of the engine's -- the parity probe wants exactly one)
enum FwdKind { FWD_REAL, FWD_BITS, FWD_IQ8, FWD_REALMIX };
static int run_forward(gpsacq_engine* e, FwdKind kind, const void* src, size_t src_stride, size_t n_src, cf* out,
                       size_t item_stride, long row, int off, bool conj_out, int sub_override = 0, const Capture* cap = nullptr) {
    const int sub = (kind == FWD_REAL || kind == FWD_REALMIX) ? 1 : (sub_override > 0 ? sub_override : e->sub);
    // k_fwd2 (the 1-bit / 8-bit IQ kernels) runs the backward transform on conjugated inputs: the conjugated spectrum is the
    // only one it can write.  Every caller wants that one (Correlate() multiplies by conj(data), :183-184; the parity probe
    // un-conjugates on the host); a future caller asking for the plain spectrum gets an error, not the wrong sign.
    if ((kind == FWD_BITS || kind == FWD_IQ8) && !conj_out)
        return fail(GPSACQ_ERR_UNSUPPORTED, "run_forward: the 1-bit / IQ forward kernels write the conjugated spectrum only");
    const size_t chunk = kFwdChunk / (size_t)sub;  // sources per launch
    for (size_t base = 0; base < n_src; base += chunk) {  // grid.y bound
        const size_t cnt = std::min(chunk, n_src - base) * (size_t)sub;
        FwdArgs fa{};
        fa.src = kind == FWD_REAL ? (const void*)((const float*)src + base * src_stride)
                 : kind == FWD_REALMIX ? (const void*)((const cf*)src + base * src_stride) : (const void*)((const uint8_t*)src + base * src_stride);
        fa.src_stride = src_stride;
        if (kind == FWD_IQ8) {
            fa.iq = cap->iq;
            fa.iq_first = cap->iq_first + base * (src_stride / 2);
            fa.iq_total = cap->iq_total;
        }
        fa.sub = sub;
        fa.lutc = e->d_lutc;
        fa.cos_t = e->d_cos_t;
        fa.sin_t = e->d_sin_t;
        fa.t1 = e->d_t1;
        fa.t2 = e->d_t2;
        fa.tn = e->d_tn;
        fa.out = out + base * (size_t)sub * item_stride;
        fa.item_stride = item_stride;
        fa.row = row;
        fa.off = off;
        fa.conj_out = conj_out ? 1 : 0;
        if (kind == FWD_BITS) launch_fwd_bits(fa, (int)cnt, e->stream);
        else if (kind == FWD_IQ8) launch_fwd_iq8(fa, (int)cnt, e->stream);
        else if (kind == FWD_REALMIX) launch_fwd_realmix(fa, (int)cnt, e->stream);
        else launch_fwd_real(fa, (int)cnt, e->stream);
    }
    HIPCHK(hipGetLastError());
    return GPSACQ_OK;
}

extern "C" const char* gpsacq_last_error(void) { return g_err.c_str(); }

extern "C" void gpsacq_destroy(gpsacq_engine* e) {
    if (!e) return;
    (void)hipSetDevice(e->p.device);
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    void* bufs[] = {e->d_t1, e->d_t2, e->d_bq, e->d_fold, e->d_tn, e->d_rho, e->d_cos, e->d_sin, e->d_cos_t, e->d_sin_t, e->d_code, e->d_patch_blocks, e->d_bits, e->d_iq, e->d_iqbits, e->d_fsamp, e->d_pdump, e->d_sums, e->d_sats, e->d_gen, e->d_persist, e->d_lutc,
                    e->d_dpp, e->d_parts, e->d_tasks, e->d_cells, e->d_peaks};
    for (void* b : bufs)
        if (b) (void)hipFree(b);
    for (auto& set : e->ev)
        for (auto& ev : set)
            if (ev) (void)hipEventDestroy(ev);
    if (e->copy_stream) (void)hipStreamSynchronize(e->copy_stream);
    for (auto& sl : e->pipe) {
        if (sl.h_in) (void)hipHostFree(sl.h_in);
        if (sl.h_peaks) (void)hipHostFree(sl.h_peaks);
        if (sl.d_in) (void)hipFree(sl.d_in);
        if (sl.d_peaks) (void)hipFree(sl.d_peaks);
        if (sl.uploaded) (void)hipEventDestroy(sl.uploaded);
        if (sl.done) (void)hipEventDestroy(sl.done);
    }
    if (e->copy_stream) (void)hipStreamDestroy(e->copy_stream);
    if (e->stream) (void)hipStreamDestroy(e->stream);
    delete e;
}

extern "C" int gpsacq_create(const gpsacq_params* params, gpsacq_engine** out) {
    if (!params || !out) return fail(GPSACQ_ERR_ARG, "gpsacq_create: null argument");
    *out = nullptr;
    if (!(params->fs > 0) || !(params->fc >= 0) || !(params->max_fo >= 0))
        return fail(GPSACQ_ERR_ARG, "gpsacq_create: need fs > 0, fc >= 0, max_fo >= 0");
    // the quadrature LO steps 4 fc / fs quadrants per sample and wraps with ONE subtraction (:155-156): at fc >= fs the
    // quadrant index leaves the 4-entry tables (undefined behaviour in the reference; rejected here)
    if (!(params->fc < params->fs)) return fail(GPSACQ_ERR_ARG, "gpsacq_create: need fc < fs (fc = %g, fs = %g)", params->fc, params->fs);
    const int dmax = doppler_half_range(params->fs, params->max_fo);
    const int nlags = num_lags(params->fs);
    const int mc = corr_columns(nlags);
    if (dmax >= N_FFT / 2) return fail(GPSACQ_ERR_UNSUPPORTED, "max_fo = %g Hz exceeds half the sampling rate", params->max_fo);

    const char* trace_env = getenv("GPSACQ_TRACE");
    const bool trace = trace_env && atoi(trace_env) != 0;
    const auto t_begin = std::chrono::steady_clock::now();
    auto lap = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count(); };
    // Everything the host computes -- twiddles, LO masks, the 32 float-sequential code replicas (15-20 ms) -- needs no device:
    // it runs in a worker while this thread sits in the HIP runtime's start-up (>100 ms in the first HIP call of a process).
    struct HostPrep {
        Tables T;
        TablesFold TF;
        std::vector<cf> tn, lutc;
        std::vector<uint8_t> cosm, sinm;
        std::vector<uint64_t> cos_t, sin_t;
        std::vector<float> rep;
        std::vector<uint32_t> chips;
    };
    const gpsacq_params prm = *params;
    auto host_prep = [prm]() {
        std::unique_ptr<HostPrep> h(new HostPrep());
        forward_tables(1, h->tn, &h->lutc);
        h->cosm.resize(BLOCK_BYTES);
        h->sinm.resize(BLOCK_BYTES);
        lo_masks(prm.fc, prm.fs, BLOCK_BYTES, h->cosm.data(), h->sinm.data());
        h->cos_t.resize(625);
        h->sin_t.resize(625);
        transpose_masks(h->cosm.data(), h->cos_t.data());
        transpose_masks(h->sinm.data(), h->sin_t.data());
        h->rep.resize((size_t)GPSACQ_NUM_SATS * N_FFT);
        for (int sv = 0; sv < GPSACQ_NUM_SATS; ++sv) code_replica(prm.fs, sv, &h->rep[(size_t)sv * N_FFT]);
        h->chips.assign(32 * 32, 0u);  // C/A chips of all 32 PRNs for the capture generator
        for (int sv = 0; sv < GPSACQ_NUM_SATS; ++sv) {
            CaCode ca(kTaps[sv][0], kTaps[sv][1]);
            for (int i = 0; i < 1023; ++i) {
                if (ca.chip()) h->chips[sv * 32 + (i >> 5)] |= 1u << (i & 31);
                ca.clock();
            }
        }
        return h;
    };
    // the worker starts BEFORE the first HIP call: the runtime's start-up is what it hides behind (an early return below waits
    // the 15-20 ms the tables take -- a delay, nothing else)
    std::future<std::unique_ptr<HostPrep>> prep_job;
    try {
        prep_job = std::async(std::launch::async, host_prep);
    } catch (const std::exception&) {  // no thread to be had (std::system_error): the tables are computed inline below; nothing crosses the C ABI
    }
    (void)hipGetLastError();  // HIP's last-error slot is sticky: do not inherit an earlier, unrelated failure of this thread
    int ndev = 0;
    hipError_t he = hipGetDeviceCount(&ndev);
    if (he != hipSuccess || ndev <= 0)
        return fail(GPSACQ_ERR_DEVICE, "no HIP device available (%s); this engine has no CPU path", hipGetErrorString(he));
    if (params->device < 0 || params->device >= ndev) return fail(GPSACQ_ERR_ARG, "device %d out of range (%d devices)", params->device, ndev);
    HIPCHK(hipSetDevice(params->device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, params->device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(GPSACQ_ERR_DEVICE, "device %d is %s; libgpsacq is built for gfx950 (MI355X) only", params->device, prop.gcnArchName);

    gpsacq_engine* e = new gpsacq_engine();
    e->p = *params;
    e->dmax = dmax;
    e->ndop = 2 * dmax + 1;
    e->dop_first = -dmax;
    e->nlags = nlags;
    e->mc = mc;
    e->kmax = dmax;
    e->halo = ((dmax + 1 + 7) / 8 + 2 + 7) & ~7;  // |floor((q - dop)/8)| <= (dmax + 1)/8 + 1 (a sub-bin grid reaches bin -(dmax + 1))
    e->crow = M_SUB + 2 * e->halo;
    e->cus = prop.multiProcessorCount;
    {
        const char* pe = getenv("GPSACQ_CORR_PERSIST");  // 0: one workgroup per cell (A/B runs)
        e->persist = !(pe && pe[0] == '0' && pe[1] == 0);  // exactly "0"
    }
    snprintf(e->name, sizeof e->name, "%s", prop.name);
#define HCK(expr)                                                                     \
    do {                                                                              \
        hipError_t e_ = (expr);                                                       \
        if (e_ != hipSuccess) {                                                       \
            int rc_ = fail(GPSACQ_ERR_DEVICE, "%s: %s", #expr, hipGetErrorString(e_)); \
            gpsacq_destroy(e);                                                        \
            return rc_;                                                               \
        }                                                                             \
    } while (0)
    const double ms_runtime = lap();  // HIP runtime + device initialisation happen inside the first calls above
    HCK(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
    for (auto& set : e->ev)
        for (auto& ev : set) HCK(hipEventCreate(&ev));

    const double ms_before_prep = lap();
    std::unique_ptr<HostPrep> hp;
    try {
        hp = prep_job.valid() ? prep_job.get() : host_prep();
    } catch (const std::exception& ex) {  // nothing may cross the C ABI as an exception
        int rc_ = fail(GPSACQ_ERR_NOMEM, "host table preparation failed: %s", ex.what());
        gpsacq_destroy(e);
        return rc_;
    }
    const Tables& T = hp->T;
    const double ms_prep_wait = lap() - ms_before_prep;
    HCK(hipMalloc((void**)&e->d_t1, T.t1.size() * sizeof(cf)));
    HCK(hipMalloc((void**)&e->d_t2, T.t2.size() * sizeof(cf)));
    HCK(hipMalloc((void**)&e->d_tn, hp->tn.size() * sizeof(cf)));
    HCK(hipMemcpy(e->d_tn, hp->tn.data(), hp->tn.size() * sizeof(cf), hipMemcpyHostToDevice));
    HCK(hipMalloc((void**)&e->d_lutc, hp->lutc.size() * sizeof(cf)));
    HCK(hipMemcpy(e->d_lutc, hp->lutc.data(), hp->lutc.size() * sizeof(cf), hipMemcpyHostToDevice));
    HCK(upload_wq(T.wq.data()));
    HCK(upload_chips(hp->chips.data()));
    {
        unsigned char rho[256] = {0};
        memcpy(rho, kRhoC, sizeof kRhoC);
        HCK(hipMalloc((void**)&e->d_rho, sizeof rho));
        HCK(hipMemcpy(e->d_rho, rho, sizeof rho, hipMemcpyHostToDevice));
    }
    HCK(hipMalloc((void**)&e->d_fold, hp->TF.fold.size() * sizeof(cf)));
    HCK(hipMemcpy(e->d_fold, hp->TF.fold.data(), hp->TF.fold.size() * sizeof(cf), hipMemcpyHostToDevice));
    HCK(hipMalloc((void**)&e->d_bq, T.bq.size() * sizeof(cf)));
    HCK(hipMemcpy(e->d_bq, T.bq.data(), T.bq.size() * sizeof(cf), hipMemcpyHostToDevice));
    HCK(hipMemcpy(e->d_t1, T.t1.data(), T.t1.size() * sizeof(cf), hipMemcpyHostToDevice));
    HCK(hipMemcpy(e->d_t2, T.t2.data(), T.t2.size() * sizeof(cf), hipMemcpyHostToDevice));

    HCK(hipMalloc((void**)&e->d_cos, BLOCK_BYTES));
    HCK(hipMalloc((void**)&e->d_sin, BLOCK_BYTES));
    HCK(hipMemcpy(e->d_cos, hp->cosm.data(), BLOCK_BYTES, hipMemcpyHostToDevice));
    HCK(hipMemcpy(e->d_sin, hp->sinm.data(), BLOCK_BYTES, hipMemcpyHostToDevice));
    HCK(hipMalloc((void**)&e->d_cos_t, 625 * sizeof(uint64_t)));
    HCK(hipMalloc((void**)&e->d_sin_t, 625 * sizeof(uint64_t)));
    HCK(hipMemcpy(e->d_cos_t, hp->cos_t.data(), 625 * sizeof(uint64_t), hipMemcpyHostToDevice));
    HCK(hipMemcpy(e->d_sin_t, hp->sin_t.data(), 625 * sizeof(uint64_t), hipMemcpyHostToDevice));

    const double ms_tables = lap();
    // SearchInit(): 32 resampled replicas (host, float-sequential NCO) -> code spectra (device)
    const std::vector<float>& rep = hp->rep;
    float* d_rep = nullptr;
    HCK(hipMalloc((void**)&d_rep, rep.size() * sizeof(float)));
    HCK(hipMemcpy(d_rep, rep.data(), rep.size() * sizeof(float), hipMemcpyHostToDevice));
    const size_t slot = (size_t)NPOLY * e->crow;
    HCK(hipMalloc((void**)&e->d_code, GPSACQ_NUM_SATS * slot * sizeof(cf)));
    HCK(hipMemsetAsync(e->d_code, 0, GPSACQ_NUM_SATS * slot * sizeof(cf), e->stream));
    {
        int rc = run_forward(e, FWD_REAL, d_rep, N_FFT, GPSACQ_NUM_SATS, e->d_code, slot, e->crow, e->halo, false);
        if (rc != GPSACQ_OK) {
            (void)hipFree(d_rep);
            gpsacq_destroy(e);
            return rc;
        }
    }
    launch_code_halo(e->d_code, GPSACQ_NUM_SATS * NPOLY, e->crow, e->halo, e->stream);
    HCK(hipGetLastError());
    HCK(hipStreamSynchronize(e->stream));
    HCK(hipFree(d_rep));
#undef HCK
    if (trace)
        fprintf(stderr, "gpsacq trace: gpsacq_create %.1f ms = HIP runtime/device start-up %.1f (host tables and code replicas computed meanwhile; "
                        "waited %.1f more for them) + stream, events, table uploads, code object load %.1f + 32 code spectra %.1f\n",
                lap(), ms_runtime, ms_prep_wait, ms_tables - ms_runtime - ms_prep_wait, lap() - ms_tables);
    *out = e;
    return GPSACQ_OK;
}

extern "C" int gpsacq_get_info(const gpsacq_engine* e, gpsacq_info* info) {
    if (!e || !info) return fail(GPSACQ_ERR_ARG, "gpsacq_get_info: null argument");
    memset(info, 0, sizeof *info);
    info->fft_len = N_FFT;
    info->dmax = e->dmax;
    info->num_doppler = e->ndop;
    info->first_doppler = e->dop_first;
    info->num_lags = e->nlags;
    info->acc_columns = e->mc;
    info->device = e->p.device;
    info->compute_units = e->cus;
    snprintf(info->device_name, sizeof info->device_name, "%s", e->name);
    info->doppler_sub = e->sub;
    info->doppler_stride = e->dstride;
    info->num_doppler_total = 2 * e->kmax + 1;
    info->first_doppler_total = -e->kmax;
    info->doppler_step_hz = e->p.fs / N_FFT * e->dstride / e->sub;
    return GPSACQ_OK;
}

// Builds the device task list.  h_tasks (host copy of the user's tasks) may be NULL for the
// reference schedule (task t = block t against PRN t % 32), whose list -- and, with ref_quirks, whose patch list
// (PRN index 0 <-> blocks 0, 32, 64, ...) -- is cached on the device: later batches of the same schedule enqueue
// nothing but the patch kernel and never wait for the stream (the pipelined front end depends on that).
static void launch_patches(gpsacq_engine* e, size_t n_patch, const uint8_t* d_bits, size_t stride) {
    QuirkArgs qa{};
    qa.code0 = e->d_code;
    qa.patched = e->d_code + (size_t)GPSACQ_NUM_SATS * NPOLY * e->crow;
    qa.bits = d_bits;
    qa.stride = stride;
    qa.block_of_patch = e->d_patch_blocks;
    qa.cos_mask = e->d_cos;
    qa.sin_mask = e->d_sin;
    qa.crow = e->crow;
    qa.halo = e->halo;
    launch_quirk_patch(qa, (int)n_patch, e->stream);
}
// The reference schedule for n_tasks tasks, uploaded and marked cached.  Every buffer it lives in (d_tasks, d_patch_blocks) is
// written here after any reallocation, and whoever else reallocates one of them (grow of d_tasks for a user task list,
// ensure_code_slots) clears sched_valid -- a cached schedule never points at memory that was not filled.
static int build_default_schedule(gpsacq_engine* e, size_t n_tasks) {
    const bool quirks = e->p.ref_quirks != 0;
    e->sched_valid = false;
    if (int rc = grow(e->d_tasks, e->task_cap, n_tasks, e->stream)) return rc;
    std::vector<Task> tk(n_tasks);
    std::vector<int32_t> patch_blocks;
    for (size_t t = 0; t < n_tasks; ++t) {
        tk[t].spec = (int32_t)t;
        tk[t].code = (int32_t)(t % GPSACQ_NUM_SATS);
        if (quirks && tk[t].code == 0) {
            tk[t].code = GPSACQ_NUM_SATS + (int32_t)patch_blocks.size();
            patch_blocks.push_back((int32_t)t);
        }
    }
    if (!patch_blocks.empty()) {
        if (int rc = ensure_code_slots(e, patch_blocks.size())) return rc;
        HIPCHK(hipMemcpyAsync(e->d_patch_blocks, patch_blocks.data(), patch_blocks.size() * sizeof(int32_t), hipMemcpyHostToDevice, e->stream));
    }
    HIPCHK(hipMemcpyAsync(e->d_tasks, tk.data(), n_tasks * sizeof(Task), hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));  // tk / patch_blocks go out of scope
    e->sched_valid = true;
    e->sched_tasks = n_tasks;
    return GPSACQ_OK;
}
static int prepare_tasks(gpsacq_engine* e, const gpsacq_task* h_tasks, const void* d_user_tasks, size_t n_blocks,
                         size_t n_tasks, const uint8_t* d_bits, size_t stride) {
    const bool quirks = e->p.ref_quirks != 0;
    const bool deflt = !h_tasks && !d_user_tasks;
    if (deflt && e->n_acc == 1) {  // cached on the device (a prefix of a longer one is the same schedule)
        if (n_tasks > n_blocks) return fail(GPSACQ_ERR_ARG, "reference schedule: %zu tasks but %zu blocks", n_tasks, n_blocks);
        if (!(e->sched_valid && n_tasks <= e->sched_tasks))
            if (int rc = build_default_schedule(e, n_tasks)) return rc;
        if (quirks) launch_patches(e, (n_tasks + GPSACQ_NUM_SATS - 1) / GPSACQ_NUM_SATS, d_bits, stride);
        return GPSACQ_OK;
    }
    e->sched_valid = false;  // before grow(): a failed reallocation must not leave a cached schedule pointing at freed memory
    if (int rc = grow(e->d_tasks, e->task_cap, n_tasks, e->stream)) return rc;
    std::vector<gpsacq_task> tmp;
    if (!h_tasks && d_user_tasks) {
        if (!quirks) {  // same layout: {block, prn} == {spec, code}
            HIPCHK(hipMemcpyAsync(e->d_tasks, d_user_tasks, n_tasks * sizeof(Task), hipMemcpyDeviceToDevice, e->stream));
            return GPSACQ_OK;
        }
        tmp.resize(n_tasks);
        HIPCHK(hipMemcpyAsync(tmp.data(), d_user_tasks, n_tasks * sizeof(Task), hipMemcpyDeviceToHost, e->stream));
        HIPCHK(hipStreamSynchronize(e->stream));
        h_tasks = tmp.data();
    }
    std::vector<Task> tk(n_tasks);
    std::vector<int32_t> patch_blocks;
    for (size_t t = 0; t < n_tasks; ++t) {
        int32_t blk = h_tasks ? h_tasks[t].block : (int32_t)t;
        int32_t prn = h_tasks ? h_tasks[t].prn : (int32_t)(t % GPSACQ_NUM_SATS);
        const size_t span = (size_t)(e->n_acc - 1) * (size_t)e->acc_step;  // last spectrum a non-coherent task touches
        if (blk < 0 || (size_t)blk + span >= n_blocks || prn < 0 || prn >= GPSACQ_NUM_SATS)
            return fail(GPSACQ_ERR_ARG, "task %zu = (block %d, prn %d) out of range (%zu blocks, %d accumulations %d apart)", t, blk, prn,
                        n_blocks, e->n_acc, e->acc_step);
        tk[t].spec = blk;
        tk[t].code = prn;
        if (quirks && prn == 0) {
            tk[t].code = GPSACQ_NUM_SATS + (int32_t)patch_blocks.size();
            patch_blocks.push_back(blk);
        }
    }
    if (!patch_blocks.empty()) {
        if (int rc = ensure_code_slots(e, patch_blocks.size())) return rc;
        HIPCHK(hipMemcpyAsync(e->d_patch_blocks, patch_blocks.data(), patch_blocks.size() * sizeof(int32_t), hipMemcpyHostToDevice, e->stream));
        launch_patches(e, patch_blocks.size(), d_bits, stride);
    }
    HIPCHK(hipMemcpyAsync(e->d_tasks, tk.data(), n_tasks * sizeof(Task), hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));  // tk / patch_blocks go out of scope
    return GPSACQ_OK;
}

static int iq8_to_bits_enqueue(gpsacq_engine* e, const uint8_t* d_iq, size_t n_samples, const IqConv& conv, size_t first_sample, uint8_t* d_bits);

// columns of pass p when the lags need several passes (fs > 10 MHz): 40 per pass, the last one the smallest instance that covers
// what is left (16.368 MHz: 66 columns = 40 + 28-column instance; round 2 ran 40 + 40)
static int pass_columns(int n_cols, int p) {
    const int left = n_cols - p * MC_MAX;
    return left >= MC_MAX ? MC_MAX : corr_columns(left * NBF3);
}

// Arms the run-time hand-out of cells for the next launch_corr (k_corr<..., PERSIST>, acq_kernels.hip): the launch then is three
// resident workgroups per CU that draw (task, Doppler point) tickets instead of one workgroup per cell.  Every instance that runs
// three workgroups per CU has a persistent form: the coherent ones and the 12-column one with its non-coherent sums in registers
// (launch_corr picks by the same conditions; for any other instance the fields are ignored).  The state is zeroed on the stream in
// front of the launch it belongs to.  GPSACQ_CORR_PERSIST=0 (read at gpsacq_create): one workgroup per cell, as up to round 5.
static int arm_persist(gpsacq_engine* e, CorrArgs& ca, int mc) {
    ca.persist_wgs = 0;
    ca.persist_queue = ca.persist_tasks = nullptr;
    const bool three_per_cu = ca.n_acc == 1 || (mc == 12 && ca.creep == 0.f && ca.lag_step == 0);
    if (!e->persist || !three_per_cu || ca.pdump || !corr_has_persistent_form(mc)) return GPSACQ_OK;
    const int wgs = 3 * e->cus;
    const Handout plan = handout_plan(ca.ndop);  // the unit of the hand-out: a task, or a ~128-point chunk of one (acq_phases.hpp)
    const int units = plan.units, chunk = plan.chunk;
    const size_t slots = (size_t)handout_slots((long)ca.n_tasks, units, wgs);
    if (slots > 0x7fffffffu / 8) return GPSACQ_OK;  // (a batch too large for the slot table: one workgroup per cell)
    const size_t ints = 9 * 16 + 8 * slots;
    if (int rc = grow(e->d_persist, e->persist_cap, ints, e->stream)) return rc;
    HIPCHK(hipMemsetAsync(e->d_persist, 0, ints * sizeof(int), e->stream));
    ca.persist_queue = e->d_persist;
    ca.persist_tasks = e->d_persist + 9 * 16;
    ca.persist_slots = (int)slots;
    ca.persist_wgs = wgs;
    ca.persist_units = units;
    ca.persist_chunk = chunk;
    return GPSACQ_OK;
}

static int search_core(gpsacq_engine* e, const Capture& cap_in, size_t n_blocks, const gpsacq_task* h_tasks,
                       const void* d_user_tasks, size_t n_tasks, Cell* d_cells, Peak* d_peaks) {
    Capture cap = cap_in;
    if (n_blocks == 0 || n_tasks == 0) return fail(GPSACQ_ERR_ARG, "empty batch");
    if (cap.iq8) {
        if (cap.stride % 16 != 0 || cap.stride < (size_t)USED_BYTES * 16) return fail(GPSACQ_ERR_ARG, "8-bit IQ blocks: stride %zu must be a multiple of 16 and >= 80000 bytes", cap.stride);
        if (((uintptr_t)cap.d_src & 15) != 0) return fail(GPSACQ_ERR_ARG, "IQ buffer must be 16-byte aligned");
        if (cap.multibit) {
            if (e->p.ref_quirks) return fail(GPSACQ_ERR_UNSUPPORTED, "ref_quirks is defined for the reference's 1-bit samples only");
        }
        if (e->p.ref_quirks) {
            // the quirk patch reads the 960 samples past each block from a 1-bit stream: convert first (same arithmetic,
            // iq_convert.hpp), then search the bits -- the un-fused route, kept for this one mode.  All 40960 samples of a
            // Sample() call are read per block: checked BEFORE the conversion is enqueued (it would read past a short upload)
            if (cap.stride < (size_t)BLOCK_BYTES * 16)
                return fail(GPSACQ_ERR_ARG, "ref_quirks needs all 40960 samples of a block: 8-bit IQ stride %zu < 81920 bytes", cap.stride);
            const size_t n_samples = (n_blocks - 1) * (cap.stride / 2) + (size_t)BLOCK_BYTES * 8;
            if (int rc = grow(e->d_iqbits, e->iqbits_cap, (n_samples + 7) / 8, e->stream)) return rc;
            IqConv cv = cap.iq;
            const size_t avail = cap.iq_total > cap.iq_first ? cap.iq_total - cap.iq_first : 0;
            if (int rc = iq8_to_bits_enqueue(e, cap.d_src, std::min(n_samples, avail), cv, cap.iq_first, e->d_iqbits)) return rc;
            cap.iq8 = false;
            cap.d_src = e->d_iqbits;
            cap.stride = cap.stride / 16;
        }
    }
    const size_t stride = cap.stride;
    if (!cap.iq8) {
        if (stride < (size_t)BLOCK_BYTES && e->p.ref_quirks) return fail(GPSACQ_ERR_ARG, "ref_quirks needs all 5120 bytes of a block (stride >= 5120)");
        if (stride < (size_t)USED_BYTES) return fail(GPSACQ_ERR_ARG, "stride %zu < 5000 bytes", stride);
    }
    if (n_blocks > 0x7fffffffu) return fail(GPSACQ_ERR_ARG, "batch too large: %zu blocks", n_blocks);
    if (n_tasks * (size_t)e->ndop > 0x7fffff00u) return fail(GPSACQ_ERR_ARG, "batch too large: %zu tasks x %d bins", n_tasks, e->ndop);
    if (n_blocks * (size_t)e->sub > 0x7fffffffu) return fail(GPSACQ_ERR_ARG, "batch too large: %zu blocks x %d sub-bin spectra", n_blocks, e->sub);
    if (int rc = grow(e->d_dpp, e->dpp_cap, n_blocks * (size_t)e->sub, e->stream, (size_t)NPOLY * M_SUB * sizeof(cf))) return rc;
    if (!d_cells) {
        if (int rc = grow(e->d_cells, e->cell_cap, n_tasks * (size_t)e->ndop, e->stream)) return rc;
        d_cells = e->d_cells;
    }
    if (int rc = prepare_tasks(e, h_tasks, d_user_tasks, n_blocks, n_tasks, cap.d_src, stride)) return rc;

    hipEvent_t* ev = e->ev[e->searches % gpsacq_engine::kTimingRing];
    HIPCHK(hipEventRecord(ev[0], e->stream));
    if (cap.iq8 && cap.multibit) {
        // multi-bit path: the real-IF value of every sample as a float (same arithmetic as the 1-bit converter, iq_convert.hpp),
        // then the forward transform with the LO applied as signs; or (multibit 2) the complex sample itself -- a capture that
        // is at baseband already, no LO
        const size_t want = (n_blocks - 1) * (stride / 2) + (size_t)USED_BYTES * 8;
        const size_t avail = cap.iq_total > cap.iq_first ? cap.iq_total - cap.iq_first : 0;
        const size_t n_samples = std::min(want, avail);
        if (n_samples < want) return fail(GPSACQ_ERR_ARG, "multi-bit path: the capture ends inside the batch (%zu of %zu samples)", n_samples, want);
        if (int rc = grow(e->d_fsamp, e->fsamp_cap, n_blocks * (size_t)e->sub * N_FFT * 2, e->stream)) return rc;  // [block][sub-bin copy][40000] complex
        IqArgs ia{};
        ia.iq = cap.d_src;
        ia.bits = nullptr;
        ia.n_samples = n_samples;
        ia.first_sample = cap.iq_first;
        ia.conv = cap.iq;
        if (cap.multibit == 2) launch_iq_to_complex(ia, stride / 2, n_blocks, e->sub, e->d_fsamp, e->stream);
        else launch_iq_to_mixed(ia, stride / 2, n_blocks, e->sub, e->d_cos, e->d_sin, e->d_fsamp, e->stream);
        HIPCHK(hipGetLastError());
        // the sub-bin copies are sources of their own here (the turn is in the samples, not in the transform's twiddles)
        if (int rc = run_forward(e, FWD_REALMIX, e->d_fsamp, N_FFT, n_blocks * (size_t)e->sub, e->d_dpp, (size_t)NPOLY * M_SUB, M_SUB, 0, true)) return rc;
    } else if (int rc = run_forward(e, cap.iq8 ? FWD_IQ8 : FWD_BITS, cap.d_src, stride, n_blocks, e->d_dpp, (size_t)NPOLY * M_SUB, M_SUB, 0, true, 0, &cap)) return rc;
    HIPCHK(hipEventRecord(ev[1], e->stream));
    // the block period in samples: what the code creeps over between accumulated blocks
    const double block_samples = cap.iq8 ? (double)stride / 2.0 : (double)stride * 8.0;
    CorrArgs ca{};
    ca.dpp = e->d_dpp;
    ca.cpp = e->d_code;
    ca.tasks = e->d_tasks;
    ca.t1 = e->d_t1;
    ca.t2 = e->d_t2;
    ca.bq = e->d_bq;
    ca.fold = e->d_fold;
    ca.rho_map = e->d_rho;
    ca.cells = d_cells;
    ca.n_tasks = (int)n_tasks;
    ca.ndop = e->ndop;
    ca.dop_first = e->dop_first;
    ca.nlags = e->nlags;
    ca.crow = e->crow;
    ca.halo = e->halo;
    ca.n_acc = e->n_acc;
    ca.acc_step = e->acc_step;
    ca.n_spec = (int)n_blocks;
    ca.creep = 0.f;
    ca.n_code = GPSACQ_NUM_SATS + (int)e->patch_cap;
    ca.sub = e->sub;
    ca.dstride = e->dstride;
    if (e->block_align && e->n_acc > 1) {
        if ((double)e->nlags * 1000.0 != e->p.fs) return fail(GPSACQ_ERR_UNSUPPORTED, "block alignment needs a whole number of samples per code period (fs = %g Hz)", e->p.fs);
        const long long t = (long long)e->acc_step * (long long)block_samples;
        ca.lag_step = (int)(t % e->nlags);
    }
    const bool realign = e->n_acc > 1 && (e->creep_comp || ca.lag_step != 0);
    const int n_cols = (e->nlags + NBF3 - 1) / NBF3;
    const int n_pass = (n_cols + MC_MAX - 1) / MC_MAX;  // 1 up to 10000 lags (fs <= 10 MHz)
    if (n_pass == 1) {
        ca.m0 = 0;
        // samples the code advances per accumulated block per Doppler bin: elapsed samples x (bin Hz / L1)
        if (e->creep_comp && e->n_acc > 1)
            ca.creep = (float)((double)e->acc_step * block_samples * (e->p.fs / N_FFT * e->dstride / e->sub) / 1575.42e6);
        if (int rc = arm_persist(e, ca, e->mc)) return rc;
        if (launch_corr(ca, e->mc, e->stream) != 0) return fail(GPSACQ_ERR_UNSUPPORTED, "no correlate kernel for %d columns", e->mc);
    } else if (realign) {
        // re-aligned lags cross the passes' column windows: the per-lag sums of every cell go to device memory (nlags floats per
        // cell), every pass adds its window's powers at their re-aligned lags, one scan per cell at the end
        // The tasks are taken in chunks so that the sums stay within kPdumpBytes (1 GB: 32 tasks x 201 points x 16368 lags are
        // 0.4 GB, a 2048-task batch would otherwise ask for 27 GB); a single task's points always go together.
        const size_t per_task = (size_t)e->ndop * (size_t)e->nlags;
        const size_t chunk_tasks = std::max<size_t>(1, std::min(n_tasks, kPdumpBytes / sizeof(float) / per_task));
        if (int rc = grow(e->d_pdump, e->pdump_cap, chunk_tasks * per_task, e->stream)) return rc;
        if (e->creep_comp) ca.creep = (float)((double)e->acc_step * block_samples * (e->p.fs / N_FFT * e->dstride / e->sub) / 1575.42e6);
        ca.pdump = e->d_pdump;
        // cells start as zeros: a task the kernel rejects (device task list out of range) leaves max_i = -1 there, which the
        // scan keeps -- the marker every other path reports
        HIPCHK(hipMemsetAsync(d_cells, 0, n_tasks * (size_t)e->ndop * sizeof(Cell), e->stream));
        for (size_t t0 = 0; t0 < n_tasks; t0 += chunk_tasks) {
            const size_t cnt = std::min(chunk_tasks, n_tasks - t0);
            HIPCHK(hipMemsetAsync(e->d_pdump, 0, cnt * per_task * sizeof(float), e->stream));
            ca.tasks = e->d_tasks + t0;
            ca.n_tasks = (int)cnt;
            ca.cells = d_cells + t0 * (size_t)e->ndop;
            for (int p = 0; p < n_pass; ++p) {
                ca.m0 = p * MC_MAX;
                const int mc = pass_columns(n_cols, p);
                if (launch_corr(ca, mc, e->stream) != 0) return fail(GPSACQ_ERR_UNSUPPORTED, "no correlate kernel for %d columns", mc);
            }
            launch_scan_power(e->d_pdump, ca.cells, cnt * (size_t)e->ndop, e->nlags, e->stream);
        }
    } else {
        const size_t n_cells = n_tasks * (size_t)e->ndop;
        if (int rc = grow(e->d_parts, e->parts_cap, n_cells * (size_t)n_pass, e->stream)) return rc;
        for (int p = 0; p < n_pass; ++p) {
            ca.m0 = p * MC_MAX;
            ca.cells = e->d_parts + (size_t)p * n_cells;
            const int mc = pass_columns(n_cols, p);
            if (int rc = arm_persist(e, ca, mc)) return rc;
            if (launch_corr(ca, mc, e->stream) != 0) return fail(GPSACQ_ERR_UNSUPPORTED, "no correlate kernel for %d columns", mc);
        }
        launch_merge_cells(e->d_parts, d_cells, n_cells, n_pass, e->nlags, e->stream);
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(ev[2], e->stream));
    launch_peaks(d_cells, d_peaks, (int)n_tasks, e->ndop, e->dop_first, e->stream);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(ev[3], e->stream));
    e->ring_launches[e->searches % gpsacq_engine::kTimingRing] = n_pass;
    e->ring_cells[e->searches % gpsacq_engine::kTimingRing] = (int64_t)n_tasks * e->ndop;
    e->searches++;
    return GPSACQ_OK;
}

static Capture bits_capture(const void* d_bits, size_t stride) {
    Capture c;
    c.d_src = (const uint8_t*)d_bits;
    c.stride = stride;
    return c;
}
// validates a gpsacq_iq8_input and turns it into the kernels' argument form
static int iq8_capture(const gpsacq_engine* e, const gpsacq_iq8_input* in, const void* d_iq, size_t stride, Capture* out) {
    if (!in) return fail(GPSACQ_ERR_ARG, "8-bit IQ search: null gpsacq_iq8_input");
    if (in->format != GPSACQ_IQ_U8 && in->format != GPSACQ_IQ_S8) return fail(GPSACQ_ERR_ARG, "unknown IQ format %d", in->format);
    if (in->multibit < 0 || in->multibit > GPSACQ_SAMPLES_COMPLEX) return fail(GPSACQ_ERR_ARG, "gpsacq_iq8_input.multibit %d: 0 (sign), 1 (real IF) or 2 (complex baseband)", in->multibit);
    const double fs = in->fs > 0 ? in->fs : e->p.fs;
    Capture c;
    c.iq8 = true;
    c.d_src = (const uint8_t*)d_iq;
    c.stride = stride;
    c.iq.is_signed = in->format == GPSACQ_IQ_S8;
    c.iq.mix = in->mix_hz != 0.0;
    c.iq.mean_i = in->remove_dc ? in->mean_i : 0.0;
    c.iq.mean_q = in->remove_dc ? in->mean_q : 0.0;
    c.iq.two_pi_fc = (2.0 * 3.141592653589793) * in->mix_hz;  // ((1i*2)*pi)*fc, left to right
    c.iq.inv_fs = 1.0 / fs;
    c.iq.fc = in->mix_hz;
    c.iq.fs = fs;
    c.iq_first = (size_t)in->first_sample;
    c.iq_total = in->total_samples ? (size_t)in->total_samples : ~(size_t)0;
    c.multibit = in->multibit;
    *out = c;
    return GPSACQ_OK;
}

extern "C" int gpsacq_search_device(gpsacq_engine* e, const void* d_bits, size_t n_blocks, size_t stride, const void* d_tasks,
                                    size_t n_tasks, void* d_cells, void* d_peaks, int sync) {
    if (!e || !d_bits || !d_peaks) return fail(GPSACQ_ERR_ARG, "gpsacq_search_device: null argument");
    if (!d_tasks && n_tasks != n_blocks && e->n_acc == 1) return fail(GPSACQ_ERR_ARG, "tasks == NULL needs n_tasks == n_blocks");
    HIPCHK(hipSetDevice(e->p.device));
    if (int rc = search_core(e, bits_capture(d_bits, stride), n_blocks, nullptr, d_tasks, n_tasks, (Cell*)d_cells, (Peak*)d_peaks)) return rc;
    if (sync) HIPCHK(hipStreamSynchronize(e->stream));
    return GPSACQ_OK;
}

extern "C" int gpsacq_search_iq8_device(gpsacq_engine* e, const gpsacq_iq8_input* in, const void* d_iq, size_t n_blocks, size_t stride,
                                        const void* d_tasks, size_t n_tasks, void* d_cells, void* d_peaks, int sync) {
    if (!e || !d_iq || !d_peaks) return fail(GPSACQ_ERR_ARG, "gpsacq_search_iq8_device: null argument");
    if (!d_tasks && n_tasks != n_blocks && e->n_acc == 1) return fail(GPSACQ_ERR_ARG, "tasks == NULL needs n_tasks == n_blocks");
    Capture cap;
    if (int rc = iq8_capture(e, in, d_iq, stride, &cap)) return rc;
    HIPCHK(hipSetDevice(e->p.device));
    if (int rc = search_core(e, cap, n_blocks, nullptr, d_tasks, n_tasks, (Cell*)d_cells, (Peak*)d_peaks)) return rc;
    if (sync) HIPCHK(hipStreamSynchronize(e->stream));
    return GPSACQ_OK;
}

// host-buffer searches: upload, search, download, wait
static int search_host(gpsacq_engine* e, Capture cap, const void* host, size_t nbytes, size_t n_blocks, const gpsacq_task* tasks,
                       size_t n_tasks, gpsacq_cell* cells, gpsacq_peak* peaks) {
    HIPCHK(hipSetDevice(e->p.device));
    uint8_t*& dbuf = cap.iq8 ? e->d_iq : e->d_bits;
    size_t& dcap = cap.iq8 ? e->iq_cap : e->bits_cap;
    if (int rc = grow(dbuf, dcap, nbytes + 16, e->stream)) return rc;
    HIPCHK(hipMemcpyAsync(dbuf, host, nbytes, hipMemcpyHostToDevice, e->stream));
    cap.d_src = dbuf;
    if (int rc = grow(e->d_cells, e->cell_cap, n_tasks * (size_t)e->ndop, e->stream)) return rc;
    if (int rc = grow(e->d_peaks, e->peak_cap, n_tasks, e->stream)) return rc;
    if (int rc = search_core(e, cap, n_blocks, tasks, nullptr, n_tasks, e->d_cells, e->d_peaks)) return rc;
    if (cells) HIPCHK(hipMemcpyAsync(cells, e->d_cells, n_tasks * (size_t)e->ndop * sizeof(Cell), hipMemcpyDeviceToHost, e->stream));
    if (peaks) HIPCHK(hipMemcpyAsync(peaks, e->d_peaks, n_tasks * sizeof(Peak), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return GPSACQ_OK;
}

extern "C" int gpsacq_search(gpsacq_engine* e, const uint8_t* bits, size_t n_blocks, size_t stride, const gpsacq_task* tasks,
                             size_t n_tasks, gpsacq_cell* cells, gpsacq_peak* peaks) {
    if (!e || !bits) return fail(GPSACQ_ERR_ARG, "gpsacq_search: null argument");
    if (!tasks && n_tasks != n_blocks && e->n_acc == 1) return fail(GPSACQ_ERR_ARG, "tasks == NULL needs n_tasks == n_blocks");
    if (n_blocks == 0 || n_tasks == 0) return fail(GPSACQ_ERR_ARG, "empty batch");
    const size_t nbytes = (n_blocks - 1) * stride + (stride < (size_t)BLOCK_BYTES ? stride : (size_t)BLOCK_BYTES);
    return search_host(e, bits_capture(nullptr, stride), bits, nbytes, n_blocks, tasks, n_tasks, cells, peaks);
}

// bytes of an 8-bit IQ buffer that hold what n_blocks blocks `stride` bytes apart read: 80000 per block, the whole 81920
// (40960 samples, one Sample() call) with ref_quirks; never beyond the end of the capture
static size_t iq8_span(const gpsacq_engine* e, const Capture& cap, size_t n_blocks) {
    const size_t per = (size_t)(e->p.ref_quirks ? BLOCK_BYTES : USED_BYTES) * 16;
    size_t nbytes = (n_blocks - 1) * cap.stride + std::min(per, cap.stride);
    if (cap.iq_total != ~(size_t)0) {
        const size_t avail = cap.iq_total > cap.iq_first ? (cap.iq_total - cap.iq_first) * 2 : 0;
        nbytes = std::min(nbytes, avail);
    }
    return nbytes;
}

extern "C" int gpsacq_search_iq8(gpsacq_engine* e, const gpsacq_iq8_input* in, const void* iq, size_t n_blocks, size_t stride,
                                 const gpsacq_task* tasks, size_t n_tasks, gpsacq_cell* cells, gpsacq_peak* peaks) {
    if (!e || !iq) return fail(GPSACQ_ERR_ARG, "gpsacq_search_iq8: null argument");
    if (!tasks && n_tasks != n_blocks && e->n_acc == 1) return fail(GPSACQ_ERR_ARG, "tasks == NULL needs n_tasks == n_blocks");
    if (n_blocks == 0 || n_tasks == 0) return fail(GPSACQ_ERR_ARG, "empty batch");
    Capture cap;
    if (int rc = iq8_capture(e, in, nullptr, stride, &cap)) return rc;
    return search_host(e, cap, iq, iq8_span(e, cap, n_blocks), n_blocks, tasks, n_tasks, cells, peaks);
}

// ---- pipelined host-buffer searches (reference schedule) -----------------------------------------------------------
// Slot life cycle: gpsacq_pipe_buffer (fill it) -> gpsacq_pipe_submit (upload on the copy stream, search on the engine's
// stream after it, peaks into pinned memory; returns at once) -> gpsacq_pipe_collect (waits for THIS slot only).  With two
// or three slots the caller's file read of batch k+1 and its report of batch k-1 overlap the search of batch k.
static int pipe_slot(gpsacq_engine* e, int slot, gpsacq_engine::PipeSlot** out) {
    if (!e) return fail(GPSACQ_ERR_ARG, "gpsacq_pipe: null engine");
    if (slot < 0 || slot >= GPSACQ_PIPE_SLOTS) return fail(GPSACQ_ERR_ARG, "gpsacq_pipe: slot %d outside 0..%d", slot, GPSACQ_PIPE_SLOTS - 1);
    HIPCHK(hipSetDevice(e->p.device));
    gpsacq_engine::PipeSlot& sl = e->pipe[slot];
    if (!e->copy_stream) HIPCHK(hipStreamCreateWithFlags(&e->copy_stream, hipStreamNonBlocking));
    if (!sl.uploaded) HIPCHK(hipEventCreateWithFlags(&sl.uploaded, hipEventDisableTiming));
    if (!sl.done) HIPCHK(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
    *out = &sl;
    return GPSACQ_OK;
}

extern "C" uint8_t* gpsacq_pipe_buffer(gpsacq_engine* e, int slot, size_t nbytes) {
    gpsacq_engine::PipeSlot* sl = nullptr;
    if (pipe_slot(e, slot, &sl) != GPSACQ_OK) return nullptr;
    if (sl->busy) {
        fail(GPSACQ_ERR_ARG, "gpsacq_pipe_buffer: slot %d has a search in flight (collect it first)", slot);
        return nullptr;
    }
    if (nbytes > sl->h_cap) {
        // everything a batch of this size needs, once: pinned staging, its device copy, and peak arrays for the most blocks
        // nbytes can hold (a block is at least 5000 bytes) -- submits then allocate nothing
        if (sl->h_in) (void)hipHostFree(sl->h_in);
        if (sl->d_in) (void)hipFree(sl->d_in);
        if (sl->d_peaks) (void)hipFree(sl->d_peaks);
        if (sl->h_peaks) (void)hipHostFree(sl->h_peaks);
        sl->h_in = nullptr;
        sl->d_in = nullptr;
        sl->d_peaks = nullptr;
        sl->h_peaks = nullptr;
        sl->h_cap = sl->d_cap = sl->peak_cap = 0;
        const size_t max_blocks = nbytes / (size_t)USED_BYTES + 1;
        hipError_t he = hipHostMalloc((void**)&sl->h_in, nbytes, hipHostMallocDefault);
        if (he == hipSuccess) he = hipMalloc((void**)&sl->d_in, nbytes + 16);
        if (he == hipSuccess) he = hipMalloc((void**)&sl->d_peaks, max_blocks * sizeof(Peak));
        if (he == hipSuccess) he = hipHostMalloc((void**)&sl->h_peaks, max_blocks * sizeof(Peak), hipHostMallocDefault);
        if (he != hipSuccess) {
            fail(he == hipErrorOutOfMemory ? GPSACQ_ERR_NOMEM : GPSACQ_ERR_DEVICE, "gpsacq_pipe_buffer(%zu bytes): %s", nbytes, hipGetErrorString(he));
            return nullptr;
        }
        sl->h_cap = nbytes;
        sl->d_cap = nbytes + 16;
        sl->peak_cap = max_blocks;
    }
    return sl->h_in;
}

extern "C" int gpsacq_pipe_submit(gpsacq_engine* e, int slot, size_t n_blocks, size_t stride, const gpsacq_iq8_input* iq) {
    gpsacq_engine::PipeSlot* sl = nullptr;
    if (int rc = pipe_slot(e, slot, &sl)) return rc;
    if (sl->busy) return fail(GPSACQ_ERR_ARG, "gpsacq_pipe_submit: slot %d already has a search in flight", slot);
    if (n_blocks == 0) return fail(GPSACQ_ERR_ARG, "empty batch");
    if (e->n_acc != 1) return fail(GPSACQ_ERR_UNSUPPORTED, "gpsacq_pipe_submit runs the reference schedule (coherent, task t = block t)");
    Capture cap = bits_capture(nullptr, stride);
    size_t nbytes;
    if (iq) {
        if (int rc = iq8_capture(e, iq, nullptr, stride, &cap)) return rc;
        nbytes = iq8_span(e, cap, n_blocks);
    } else {
        nbytes = (n_blocks - 1) * stride + (stride < (size_t)BLOCK_BYTES ? stride : (size_t)BLOCK_BYTES);
    }
    if (!sl->h_in || nbytes > sl->h_cap) return fail(GPSACQ_ERR_ARG, "gpsacq_pipe_submit: slot %d holds %zu bytes, the batch needs %zu", slot, sl->h_cap, nbytes);
    if (nbytes + 16 > sl->d_cap || n_blocks > sl->peak_cap)
        return fail(GPSACQ_ERR_ARG, "gpsacq_pipe_submit: slot %d was sized for %zu bytes / %zu blocks", slot, sl->h_cap, sl->peak_cap);
    HIPCHK(hipMemcpyAsync(sl->d_in, sl->h_in, nbytes, hipMemcpyHostToDevice, e->copy_stream));
    HIPCHK(hipEventRecord(sl->uploaded, e->copy_stream));
    HIPCHK(hipStreamWaitEvent(e->stream, sl->uploaded, 0));
    cap.d_src = sl->d_in;
    if (int rc = search_core(e, cap, n_blocks, nullptr, nullptr, n_blocks, nullptr, sl->d_peaks)) return rc;
    HIPCHK(hipMemcpyAsync(sl->h_peaks, sl->d_peaks, n_blocks * sizeof(Peak), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipEventRecord(sl->done, e->stream));
    sl->busy = true;
    sl->n_tasks = n_blocks;
    return GPSACQ_OK;
}

extern "C" int gpsacq_pipe_collect(gpsacq_engine* e, int slot, const gpsacq_peak** peaks, size_t* n_peaks) {
    gpsacq_engine::PipeSlot* sl = nullptr;
    if (int rc = pipe_slot(e, slot, &sl)) return rc;
    if (!sl->busy) return fail(GPSACQ_ERR_ARG, "gpsacq_pipe_collect: slot %d has nothing in flight", slot);
    HIPCHK(hipEventSynchronize(sl->done));
    sl->busy = false;
    if (peaks) *peaks = reinterpret_cast<const gpsacq_peak*>(sl->h_peaks);
    if (n_peaks) *n_peaks = sl->n_tasks;
    return GPSACQ_OK;
}

// Scratch for batches of up to n_blocks blocks (reference schedule), allocated once: a caller that knows its largest batch --
// the pipelined front end -- spares every later search the stream-ordered regrowth of the spectra / cell / task buffers.
extern "C" int gpsacq_reserve(gpsacq_engine* e, size_t n_blocks) {
    if (!e || n_blocks == 0) return fail(GPSACQ_ERR_ARG, "gpsacq_reserve: bad argument");
    HIPCHK(hipSetDevice(e->p.device));
    if (int rc = grow(e->d_dpp, e->dpp_cap, n_blocks * (size_t)e->sub, e->stream, (size_t)NPOLY * M_SUB * sizeof(cf))) return rc;
    if (int rc = grow(e->d_cells, e->cell_cap, n_blocks * (size_t)e->ndop, e->stream)) return rc;
    if (e->n_acc == 1) {
        // the reference schedule for the largest batch, built once: every shorter batch (the front end's 1, 2, 4, ... run ramp)
        // is a prefix of it and enqueues without waiting for the stream
        if (!(e->sched_valid && n_blocks <= e->sched_tasks))
            if (int rc = build_default_schedule(e, n_blocks)) return rc;
    } else {
        e->sched_valid = false;  // (never valid in the non-coherent mode; the buffer may be about to move)
        if (int rc = grow(e->d_tasks, e->task_cap, n_blocks, e->stream)) return rc;
    }
    return GPSACQ_OK;
}

extern "C" int gpsacq_set_noncoherent(gpsacq_engine* e, int n_acc, int block_step) {
    if (!e) return fail(GPSACQ_ERR_ARG, "gpsacq_set_noncoherent: null engine");
    if (n_acc < 1 || n_acc > 1024 || (n_acc > 1 && block_step < 1)) return fail(GPSACQ_ERR_ARG, "need 1 <= n_acc <= 1024 and block_step >= 1");
    if (n_acc > 1 && e->p.ref_quirks) return fail(GPSACQ_ERR_UNSUPPORTED, "ref_quirks is defined for the reference's coherent search only");
    e->n_acc = n_acc;
    e->acc_step = n_acc > 1 ? block_step : 0;
    e->sched_valid = false;
    return GPSACQ_OK;
}

extern "C" int gpsacq_set_creep_compensation(gpsacq_engine* e, int on) {
    if (!e) return fail(GPSACQ_ERR_ARG, "gpsacq_set_creep_compensation: null engine");
    e->creep_comp = on != 0;
    return GPSACQ_OK;
}

extern "C" int gpsacq_set_block_alignment(gpsacq_engine* e, int on) {
    if (!e) return fail(GPSACQ_ERR_ARG, "gpsacq_set_block_alignment: null engine");
    e->block_align = on != 0;
    return GPSACQ_OK;
}

extern "C" int gpsacq_aligned_stride(const gpsacq_engine* e) {
    if (!e) return -1;
    if (e->nlags % 8 != 0) return -1;  // a code period is not a whole number of bytes
    const int per = e->nlags / 8;      // bytes per code period (FS/1000 samples)
    return per * ((BLOCK_BYTES + per - 1) / per);
}

extern "C" int gpsacq_set_doppler_window(gpsacq_engine* e, int first_bin, int n_bins) {
    if (!e) return fail(GPSACQ_ERR_ARG, "gpsacq_set_doppler_window: null engine");
    if (n_bins <= 0 || first_bin < -e->kmax || first_bin + n_bins - 1 > e->kmax)
        return fail(GPSACQ_ERR_ARG, "Doppler window [%d, %d] outside [-%d, %d]", first_bin, first_bin + n_bins - 1, e->kmax, e->kmax);
    e->dop_first = first_bin;
    e->ndop = n_bins;
    return GPSACQ_OK;
}

extern "C" int gpsacq_set_cell_handout(gpsacq_engine* e, int on) {
    if (!e) return fail(GPSACQ_ERR_ARG, "gpsacq_set_cell_handout: null engine");
    e->persist = on != 0;
    return GPSACQ_OK;
}

extern "C" int gpsacq_set_doppler_step(gpsacq_engine* e, double step_hz) {
    if (!e) return fail(GPSACQ_ERR_ARG, "gpsacq_set_doppler_step: null engine");
    HIPCHK(hipSetDevice(e->p.device));
    const double bin = e->p.fs / N_FFT;
    int sub = 1, dstride = 1;
    if (step_hz > 0 && step_hz < bin * (1 - 1e-9)) sub = (int)std::ceil(bin / step_hz - 1e-9);
    else if (step_hz >= 2 * bin * (1 - 1e-9)) dstride = (int)std::floor(step_hz / bin + 1e-9);
    if (sub > GPSACQ_MAX_DOPPLER_SUB) return fail(GPSACQ_ERR_UNSUPPORTED, "Doppler step %g Hz needs %d sub-bin spectra per block (limit %d)", step_hz, sub, GPSACQ_MAX_DOPPLER_SUB);
    if (sub > 1 && e->p.ref_quirks) return fail(GPSACQ_ERR_UNSUPPORTED, "ref_quirks is defined for the reference's Doppler grid only");
    // same truncation and operation order as :176, (int)(max_fo * N / fs), with the step folded in; sub = dstride = 1
    // (step_hz = 0 restores the reference grid) gives exactly the dmax of gpsacq_create
    const int kmax = (sub == 1 && dstride == 1) ? doppler_half_range(e->p.fs, e->p.max_fo)
                                                : (int)(e->p.max_fo * (double)N_FFT * (double)sub / (e->p.fs * (double)dstride));
    if (2 * kmax + 1 > 0xFFFF) return fail(GPSACQ_ERR_UNSUPPORTED, "%d Doppler points exceed the 65535 the peak keys can carry", 2 * kmax + 1);
    if (sub != e->sub) {  // forward-transform tables of the sub-bin offsets
        HIPCHK(hipStreamSynchronize(e->stream));
        std::vector<cf> tn, lutc;
        forward_tables(sub, tn, &lutc);
        cf *ntn = nullptr, *nlut = nullptr;
        HIPCHK(hipMalloc((void**)&ntn, tn.size() * sizeof(cf)));
        HIPCHK(hipMalloc((void**)&nlut, lutc.size() * sizeof(cf)));
        HIPCHK(hipMemcpy(ntn, tn.data(), tn.size() * sizeof(cf), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(nlut, lutc.data(), lutc.size() * sizeof(cf), hipMemcpyHostToDevice));
        HIPCHK(hipFree(e->d_tn));
        HIPCHK(hipFree(e->d_lutc));
        e->d_tn = ntn;
        e->d_lutc = nlut;
    }
    e->sub = sub;
    e->dstride = dstride;
    e->kmax = kmax;
    e->dop_first = -kmax;
    e->ndop = 2 * kmax + 1;
    return GPSACQ_OK;
}

extern "C" int gpsacq_synchronize(gpsacq_engine* e) {
    if (!e) return fail(GPSACQ_ERR_ARG, "gpsacq_synchronize: null engine");
    HIPCHK(hipSetDevice(e->p.device));
    HIPCHK(hipStreamSynchronize(e->stream));
    return GPSACQ_OK;
}

// Merge keys of a finished (or enqueued) device search, on the engine's stream: bench.py's / a multi-process caller's step is
// then one search + this + one all-reduce(MAX), with nothing of it on another stream or in another framework's kernels.
extern "C" int gpsacq_peak_keys_device(gpsacq_engine* e, const void* d_peaks, size_t n_peaks, int per_prn, void* d_keys, int sync) {
    if (!e || !d_keys || (!d_peaks && n_peaks > 0)) return fail(GPSACQ_ERR_ARG, "gpsacq_peak_keys_device: null argument");
    if (n_peaks > 0x7fffffffu) return fail(GPSACQ_ERR_ARG, "gpsacq_peak_keys_device: %zu peaks", n_peaks);
    HIPCHK(hipSetDevice(e->p.device));
    if (per_prn) launch_prn_keys((const Peak*)d_peaks, (int)n_peaks, e->kmax, (unsigned long long*)d_keys, e->stream);
    else if (n_peaks > 0) launch_pack_keys((const Peak*)d_peaks, (unsigned long long*)d_keys, (int)n_peaks, e->kmax, e->stream);
    HIPCHK(hipGetLastError());
    if (sync) HIPCHK(hipStreamSynchronize(e->stream));
    return GPSACQ_OK;
}

extern "C" int gpsacq_cycle_stamp_device(gpsacq_engine* e, void* d_stamp, int sync) {
    if (!e || !d_stamp) return fail(GPSACQ_ERR_ARG, "gpsacq_cycle_stamp_device: null argument");
    HIPCHK(hipSetDevice(e->p.device));
    launch_cycle_stamp((unsigned long long*)d_stamp, e->stream);
    HIPCHK(hipGetLastError());
    if (sync) HIPCHK(hipStreamSynchronize(e->stream));
    return GPSACQ_OK;
}

extern "C" int gpsacq_timing_ago(const gpsacq_engine* e, int n_back, gpsacq_timing* t) {
    if (!e || !t) return fail(GPSACQ_ERR_ARG, "gpsacq_timing_ago: null argument");
    if (n_back < 0 || n_back >= gpsacq_engine::kTimingRing || (long)n_back >= e->searches)
        return fail(GPSACQ_ERR_ARG, "no timing %d searches back (%ld searches so far, %d kept)", n_back, e->searches, gpsacq_engine::kTimingRing);
    const int slot = (int)((e->searches - 1 - n_back) % gpsacq_engine::kTimingRing);
    const hipEvent_t* ev = e->ev[slot];
    HIPCHK(hipEventSynchronize(ev[3]));
    memset(t, 0, sizeof *t);
    HIPCHK(hipEventElapsedTime(&t->ms_total, ev[0], ev[3]));
    HIPCHK(hipEventElapsedTime(&t->ms_sample, ev[0], ev[1]));
    HIPCHK(hipEventElapsedTime(&t->ms_correlate, ev[1], ev[2]));
    HIPCHK(hipEventElapsedTime(&t->ms_peaks, ev[2], ev[3]));
    t->correlate_launches = e->ring_launches[slot];
    t->cells = e->ring_cells[slot];
    return GPSACQ_OK;
}

extern "C" int gpsacq_last_timing(const gpsacq_engine* e, gpsacq_timing* t) { return gpsacq_timing_ago(e, 0, t); }

extern "C" void* gpsacq_stream(gpsacq_engine* e) { return e ? (void*)e->stream : nullptr; }

// Synthetic capture on the device (gps_sig_gen.m's role; signal model of SURVEY.md section 8d).  The stream is a function of
// (seed, satellites, absolute sample index) alone: any byte range of it can be generated anywhere -- a rank of a multi-GPU job
// makes exactly its own blocks of the one capture every world size searches.
extern "C" int gpsacq_generate_range_device(gpsacq_engine* e, void* d_bits, size_t n_bytes, uint64_t first_sample, const gpsacq_sat* sats,
                                            int n_sats, float noise_sigma, uint64_t seed, int sync) {
    if (!e || !d_bits || n_bytes == 0 || n_sats < 0 || (n_sats > 0 && !sats)) return fail(GPSACQ_ERR_ARG, "gpsacq_generate_range_device: bad argument");
    if (first_sample & 7) return fail(GPSACQ_ERR_ARG, "gpsacq_generate_range_device: first_sample %llu is not a multiple of 8 (a byte boundary)", (unsigned long long)first_sample);
    HIPCHK(hipSetDevice(e->p.device));
    const double L1 = 1575.42e6, CPS = 1.023e6;
    std::vector<GenSat> gs((size_t)n_sats);
    for (int i = 0; i < n_sats; ++i) {
        if (sats[i].prn < 1 || sats[i].prn > GPSACQ_NUM_SATS) return fail(GPSACQ_ERR_ARG, "satellite %d: PRN %d out of 1..32", i, sats[i].prn);
        gs[i].sv = sats[i].prn - 1;
        gs[i].amplitude = sats[i].amplitude;
        gs[i].chips_per_sample = CPS * (1.0 + sats[i].doppler_hz / L1) / e->p.fs;
        gs[i].code_phase = sats[i].code_phase_samples;
        gs[i].cycles_per_sample = (e->p.fc + sats[i].doppler_hz) / e->p.fs;
        gs[i].carrier_phase = sats[i].carrier_phase_cycles;
    }
    if (n_sats > 0) {
        if (int rc = grow(e->d_sats, e->sats_cap, (size_t)n_sats, e->stream)) return rc;
        HIPCHK(hipMemcpyAsync(e->d_sats, gs.data(), gs.size() * sizeof(GenSat), hipMemcpyHostToDevice, e->stream));
        HIPCHK(hipStreamSynchronize(e->stream));  // gs goes out of scope
    }
    GenArgs a{};
    a.bits = (uint8_t*)d_bits;
    a.n_bytes = n_bytes;
    a.first_sample = first_sample;
    a.seed = seed;
    a.sats = e->d_sats;
    a.n_sats = n_sats;
    a.noise_sigma = noise_sigma;
    launch_generate(a, e->stream);
    HIPCHK(hipGetLastError());
    if (sync) HIPCHK(hipStreamSynchronize(e->stream));
    return GPSACQ_OK;
}
extern "C" int gpsacq_generate_device(gpsacq_engine* e, void* d_bits, size_t n_bytes, const gpsacq_sat* sats, int n_sats,
                                      float noise_sigma, uint64_t seed, int sync) {
    return gpsacq_generate_range_device(e, d_bits, n_bytes, 0, sats, n_sats, noise_sigma, seed, sync);
}

extern "C" int gpsacq_generate_range(gpsacq_engine* e, uint8_t* bits_out, size_t n_bytes, uint64_t first_sample, const gpsacq_sat* sats,
                                     int n_sats, float noise_sigma, uint64_t seed) {
    if (!e || !bits_out || n_bytes == 0) return fail(GPSACQ_ERR_ARG, "gpsacq_generate_range: bad argument");
    HIPCHK(hipSetDevice(e->p.device));
    if (int rc = grow(e->d_gen, e->gen_cap, n_bytes, e->stream)) return rc;
    if (int rc = gpsacq_generate_range_device(e, e->d_gen, n_bytes, first_sample, sats, n_sats, noise_sigma, seed, 0)) return rc;
    HIPCHK(hipMemcpyAsync(bits_out, e->d_gen, n_bytes, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return GPSACQ_OK;
}
extern "C" int gpsacq_generate(gpsacq_engine* e, uint8_t* bits_out, size_t n_bytes, const gpsacq_sat* sats, int n_sats,
                               float noise_sigma, uint64_t seed) {
    return gpsacq_generate_range(e, bits_out, n_bytes, 0, sats, n_sats, noise_sigma, seed);
}

// gps_sig_gen.m's signal for any PRN / navigation bits (k_siggen)
extern "C" size_t gpsacq_sig_bytes(int n_data_bits) {
    if (n_data_bits < 1) return 0;
    const long long n = (long long)n_data_bits * 20 * 1023 * 8 + 48;
    return (size_t)((n + 7) / 8);
}
extern "C" int gpsacq_generate_sig(gpsacq_engine* e, int prn, const int8_t* data_bits, int n_data_bits, uint8_t* bits_out, size_t n_bytes) {
    if (!e || !data_bits || !bits_out || n_data_bits < 1 || n_data_bits > 100000) return fail(GPSACQ_ERR_ARG, "gpsacq_generate_sig: bad argument");
    if (prn < 1 || prn > GPSACQ_NUM_SATS) return fail(GPSACQ_ERR_ARG, "gpsacq_generate_sig: PRN %d out of 1..32", prn);
    if (n_bytes != gpsacq_sig_bytes(n_data_bits)) return fail(GPSACQ_ERR_ARG, "gpsacq_generate_sig: %d bits make %zu bytes, not %zu", n_data_bits, gpsacq_sig_bytes(n_data_bits), n_bytes);
    for (int i = 0; i < n_data_bits; ++i)
        if (data_bits[i] != 1 && data_bits[i] != -1) return fail(GPSACQ_ERR_ARG, "gpsacq_generate_sig: navigation bit %d is %d, not +-1", i, data_bits[i]);
    HIPCHK(hipSetDevice(e->p.device));
    if (int rc = grow(e->d_gen, e->gen_cap, n_bytes + (size_t)n_data_bits, e->stream)) return rc;
    int8_t* d_data = (int8_t*)(e->d_gen + n_bytes);
    HIPCHK(hipMemcpyAsync(d_data, data_bits, (size_t)n_data_bits, hipMemcpyHostToDevice, e->stream));
    const double ca_rate = 1.023e6 * 8, fc = ca_rate / 4;  // gps_sig_gen.m:8-9,14,34
    SigArgs a{};
    a.bits = e->d_gen;
    a.n_bytes = n_bytes;
    a.n_samples = (long long)n_data_bits * 20 * 1023 * 8 + 48;
    a.data = d_data;
    a.n_data = n_data_bits;
    a.sv = prn - 1;
    a.two_pi_fc = (2.0 * 3.141592653589793) * fc;  // 1i.*2.*pi.*fc, left to right (:36)
    a.inv_rate = 1.0 / ca_rate;
    launch_siggen(a, e->stream);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(bits_out, e->d_gen, n_bytes, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return GPSACQ_OK;
}

// gps_sig_gen.m:21-30: the script's HackRF transmit file, any range of its samples (k_siggen_tx)
extern "C" uint64_t gpsacq_sig_tx_samples(int n_data_bits, int n_repeat) {
    if (n_data_bits < 1 || n_repeat < 1) return 0;
    return (uint64_t)n_repeat * (uint64_t)n_data_bits * 20 * 1023 * 8 + 48;
}
extern "C" int gpsacq_generate_sig_tx(gpsacq_engine* e, int prn, const int8_t* data_bits, int n_data_bits, int n_repeat,
                                      uint64_t first_sample, size_t n_samples, int8_t* iq_out) {
    if (!e || !data_bits || !iq_out || n_data_bits < 1 || n_data_bits > 100000 || n_repeat < 1 || n_repeat > 1000 || n_samples == 0)
        return fail(GPSACQ_ERR_ARG, "gpsacq_generate_sig_tx: bad argument");
    if (prn < 1 || prn > GPSACQ_NUM_SATS) return fail(GPSACQ_ERR_ARG, "gpsacq_generate_sig_tx: PRN %d out of 1..32", prn);
    const uint64_t total = gpsacq_sig_tx_samples(n_data_bits, n_repeat);
    if (first_sample > total || n_samples > total - first_sample)
        return fail(GPSACQ_ERR_ARG, "gpsacq_generate_sig_tx: samples %llu..%llu outside the stream of %llu", (unsigned long long)first_sample,
                    (unsigned long long)(first_sample + n_samples), (unsigned long long)total);
    for (int i = 0; i < n_data_bits; ++i)
        if (data_bits[i] != 1 && data_bits[i] != -1) return fail(GPSACQ_ERR_ARG, "gpsacq_generate_sig_tx: navigation bit %d is %d, not +-1", i, data_bits[i]);
    HIPCHK(hipSetDevice(e->p.device));
    const size_t out_bytes = 2 * n_samples, data_off = (out_bytes + 15) & ~(size_t)15;
    if (int rc = grow(e->d_gen, e->gen_cap, data_off + (size_t)n_data_bits, e->stream)) return rc;
    int8_t* d_data = (int8_t*)(e->d_gen + data_off);
    HIPCHK(hipMemcpyAsync(d_data, data_bits, (size_t)n_data_bits, hipMemcpyHostToDevice, e->stream));
    SigTxArgs a{};
    a.iq = (int8_t*)e->d_gen;
    a.n_samples = n_samples;
    a.first_sample = (long long)first_sample;
    a.data = d_data;
    a.n_data = n_data_bits;
    a.n_repeat = n_repeat;
    a.sv = prn - 1;
    launch_siggen_tx(a, e->stream);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(iq_out, e->d_gen, out_bytes, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return GPSACQ_OK;
}

// 8-bit IQ -> 1-bit real IF (proc_rtl_bin_for_gps.m / proc_hackrf_bin_for_gps.m), device buffers
static int iq8_to_bits_enqueue(gpsacq_engine* e, const uint8_t* d_iq, size_t n_samples, const IqConv& conv, size_t first_sample, uint8_t* d_bits) {
    if (n_samples == 0) return GPSACQ_OK;
    IqArgs a{};
    a.iq = d_iq;
    a.bits = d_bits;
    a.n_samples = n_samples;
    a.first_sample = first_sample;
    a.conv = conv;
    launch_iq_to_bits(a, e->stream);
    HIPCHK(hipGetLastError());
    return GPSACQ_OK;
}

// exact integer sums of I and Q over n_samples of a device buffer, added to sums[0..1]
static int iq8_sums_device(gpsacq_engine* e, const uint8_t* d_iq, size_t n_samples, bool is_signed, long long sums[2]) {
    if (!e->d_sums) HIPCHK(hipMalloc((void**)&e->d_sums, 2 * sizeof(unsigned long long)));
    HIPCHK(hipMemsetAsync(e->d_sums, 0, 2 * sizeof(unsigned long long), e->stream));
    launch_iq_sums(d_iq, n_samples, is_signed, e->d_sums, e->stream);
    HIPCHK(hipGetLastError());
    long long h[2];
    HIPCHK(hipMemcpyAsync(h, e->d_sums, sizeof h, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    sums[0] += h[0];
    sums[1] += h[1];
    return GPSACQ_OK;
}

extern "C" int gpsacq_iq8_to_bits_device(gpsacq_engine* e, const void* d_iq, size_t n_samples, int format, int remove_dc,
                                         double mix_hz, double fs, void* d_bits, int sync) {
    if (!e || !d_iq || !d_bits || n_samples == 0) return fail(GPSACQ_ERR_ARG, "gpsacq_iq8_to_bits_device: bad argument");
    if (format != GPSACQ_IQ_U8 && format != GPSACQ_IQ_S8) return fail(GPSACQ_ERR_ARG, "unknown IQ format %d", format);
    if (((uintptr_t)d_iq & 15) != 0) return fail(GPSACQ_ERR_ARG, "IQ buffer must be 16-byte aligned");
    if (!(fs > 0)) fs = e->p.fs;
    HIPCHK(hipSetDevice(e->p.device));
    IqConv c{};
    c.is_signed = format == GPSACQ_IQ_S8;
    c.mix = mix_hz != 0.0;
    c.two_pi_fc = (2.0 * 3.141592653589793) * mix_hz;  // ((1i*2)*pi)*fc, left to right
    c.inv_fs = 1.0 / fs;
    c.fc = mix_hz;
    c.fs = fs;
    if (remove_dc) {
        long long h[2] = {0, 0};
        if (int rc = iq8_sums_device(e, (const uint8_t*)d_iq, n_samples, c.is_signed != 0, h)) return rc;
        c.mean_i = (double)h[0] / (double)n_samples;  // mean() of integer-valued doubles: exact sums
        c.mean_q = (double)h[1] / (double)n_samples;
    }
    if (int rc = iq8_to_bits_enqueue(e, (const uint8_t*)d_iq, n_samples, c, 0, (uint8_t*)d_bits)) return rc;
    if (sync) HIPCHK(hipStreamSynchronize(e->stream));
    return GPSACQ_OK;
}

// `y - mean(y)` is over the WHOLE capture (proc_rtl_bin_for_gps.m:17): a caller that streams a file in pieces sums first
extern "C" int gpsacq_iq8_accumulate_sums(gpsacq_engine* e, const void* iq, size_t n_samples, int format, int64_t sums[2]) {
    if (!e || !iq || !sums || n_samples == 0) return fail(GPSACQ_ERR_ARG, "gpsacq_iq8_accumulate_sums: bad argument");
    if (format != GPSACQ_IQ_U8 && format != GPSACQ_IQ_S8) return fail(GPSACQ_ERR_ARG, "unknown IQ format %d", format);
    HIPCHK(hipSetDevice(e->p.device));
    if (int rc = grow(e->d_iq, e->iq_cap, 2 * n_samples + 16, e->stream)) return rc;
    HIPCHK(hipMemcpyAsync(e->d_iq, iq, 2 * n_samples, hipMemcpyHostToDevice, e->stream));
    long long h[2] = {0, 0};
    if (int rc = iq8_sums_device(e, e->d_iq, n_samples, format == GPSACQ_IQ_S8, h)) return rc;
    sums[0] += h[0];
    sums[1] += h[1];
    return GPSACQ_OK;
}

extern "C" int gpsacq_iq8_to_bits(gpsacq_engine* e, const void* iq, size_t n_samples, int format, int remove_dc, double mix_hz,
                                  double fs, uint8_t* bits_out) {
    if (!e || !iq || !bits_out || n_samples == 0) return fail(GPSACQ_ERR_ARG, "gpsacq_iq8_to_bits: bad argument");
    HIPCHK(hipSetDevice(e->p.device));
    const size_t n_bytes = (n_samples + 7) / 8;
    if (int rc = grow(e->d_iq, e->iq_cap, 2 * n_samples + 16, e->stream)) return rc;
    if (int rc = grow(e->d_iqbits, e->iqbits_cap, n_bytes, e->stream)) return rc;
    HIPCHK(hipMemcpyAsync(e->d_iq, iq, 2 * n_samples, hipMemcpyHostToDevice, e->stream));
    if (int rc = gpsacq_iq8_to_bits_device(e, e->d_iq, n_samples, format, remove_dc, mix_hz, fs, e->d_iqbits, 0)) return rc;
    HIPCHK(hipMemcpyAsync(bits_out, e->d_iqbits, n_bytes, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return GPSACQ_OK;
}

extern "C" int gpsacq_handoff_step(const gpsacq_peak* peak, double fc, double fs, double step_hz, double secs, gpsacq_handoff_t* out) {
    if (!peak || !out || !(fs > 0)) return fail(GPSACQ_ERR_ARG, "gpsacq_handoff: bad argument");
    const double L1 = 1575.42e6, CPS = 1.023e6;  // c/gps_offline.h:22,30
    // lo_shift counts Doppler grid points: FFT bins of fs / 40000 Hz on the reference grid (:145), gpsacq_info.doppler_step_hz
    // after gpsacq_set_doppler_step and for gpsacq_multi_search_grid peaks
    const double lo_dop = step_hz > 0 ? peak->lo_shift * step_hz : peak->lo_shift * fs / N_FFT;
    const double ca_dop = lo_dop / L1 * CPS;
    // the NCO words are fractions of 2^32: frequencies outside [0, fs) do not fit (the reference would wrap silently)
    if (!(fc + lo_dop >= 0 && fc + lo_dop < fs) || !(CPS + ca_dop >= 0 && CPS + ca_dop < fs))
        return fail(GPSACQ_ERR_ARG, "gpsacq_handoff: carrier %g Hz or code rate %g Hz outside [0, fs = %g)", fc + lo_dop, CPS + ca_dop, fs);
    out->lo_dop_hz = lo_dop;
    out->ca_dop_hz = ca_dop;
    out->lo_rate = (uint32_t)((fc + lo_dop) / fs * 4294967296.0);
    out->ca_rate = (uint32_t)((CPS + ca_dop) / fs * 4294967296.0);
    const int spm = num_lags(fs);
    int ca = peak->ca_shift + (int)nearbyint(ca_dop * secs * fs / CPS);
    out->ca_shift = ca;
    int pause = (2 * spm - ca) % spm;
    if (pause < 0) pause += spm;
    out->ca_pause = (uint32_t)pause;
    return GPSACQ_OK;
}

extern "C" int gpsacq_handoff(const gpsacq_peak* peak, double fc, double fs, double secs, gpsacq_handoff_t* out) {
    return gpsacq_handoff_step(peak, fc, fs, 0.0, secs, out);  // the reference grid: lo_shift in FFT bins
}

extern "C" int gpsacq_handoff_engine(const gpsacq_engine* e, const gpsacq_peak* peak, double secs, gpsacq_handoff_t* out) {
    if (!e) return fail(GPSACQ_ERR_ARG, "gpsacq_handoff_engine: null engine");
    const bool ref_grid = e->sub == 1 && e->dstride == 1;
    return gpsacq_handoff_step(peak, e->p.fc, e->p.fs, ref_grid ? 0.0 : e->p.fs / N_FFT * e->dstride / e->sub, secs, out);
}

extern "C" int gpsacq_search_code(int sv, int g1) {
    if (sv < 0 || sv >= GPSACQ_NUM_SATS) return -1;
    return search_code(sv, g1);
}

static void pp_to_natural(const std::vector<cf>& pp, long row, int off, bool conj, float* out) {
    for (int k = 0; k < N_FFT; ++k) {
        const cf v = pp[(size_t)(k & 7) * row + off + (k >> 3)];
        out[2 * k] = v.x;
        out[2 * k + 1] = conj ? -v.y : v.y;
    }
}

extern "C" int gpsacq_sample_spectrum(gpsacq_engine* e, const uint8_t* block, float* out) {
    if (!e || !block || !out) return fail(GPSACQ_ERR_ARG, "gpsacq_sample_spectrum: null argument");
    HIPCHK(hipSetDevice(e->p.device));
    if (int rc = grow(e->d_bits, e->bits_cap, (size_t)BLOCK_BYTES, e->stream)) return rc;
    if (int rc = grow(e->d_dpp, e->dpp_cap, (size_t)1, e->stream, (size_t)NPOLY * M_SUB * sizeof(cf))) return rc;
    HIPCHK(hipMemcpyAsync(e->d_bits, block, BLOCK_BYTES, hipMemcpyHostToDevice, e->stream));
    // one spectrum (sub-bin offset 0) even when a finer Doppler grid is active: d_dpp is grown for exactly that
    if (int rc = run_forward(e, FWD_BITS, e->d_bits, BLOCK_BYTES, 1, e->d_dpp, (size_t)NPOLY * M_SUB, M_SUB, 0, true, 1)) return rc;
    std::vector<cf> pp((size_t)NPOLY * M_SUB);
    HIPCHK(hipMemcpyAsync(pp.data(), e->d_dpp, pp.size() * sizeof(cf), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    pp_to_natural(pp, M_SUB, 0, true, out);  // stored conjugated; return the spectrum itself
    return GPSACQ_OK;
}

extern "C" int gpsacq_code_spectrum(gpsacq_engine* e, int sv, float* out) {
    if (!e || !out || sv < 0 || sv >= GPSACQ_NUM_SATS) return fail(GPSACQ_ERR_ARG, "gpsacq_code_spectrum: bad argument");
    HIPCHK(hipSetDevice(e->p.device));
    std::vector<cf> pp((size_t)NPOLY * e->crow);
    HIPCHK(hipMemcpy(pp.data(), e->d_code + (size_t)sv * NPOLY * e->crow, pp.size() * sizeof(cf), hipMemcpyDeviceToHost));
    pp_to_natural(pp, e->crow, e->halo, false, out);
    return GPSACQ_OK;
}
