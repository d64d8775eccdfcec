// gpsacq_multi.cpp -- single-process multi-GPU search of one capture's PRN x Doppler grid (include/gpsacq.h,
// gpsacq_multi_*): one engine per device, the Doppler grid cut into one contiguous slab per device, and ONE
// ncclAllReduce(MAX) of the packed per-task peak keys over RCCL (xGMI on an 8 x MI355X node) -- the only exchange
// step of the path (BASELINE.json north_star: "RCCL all-reduce of the per-PRN peak only").  The reference has no
// parallelism at all (c/search_offline.cpp is one thread); every (block, PRN, Doppler) cell is independent.
//
// Engines may share a physical GPU (the same ordinal listed more than once): RCCL joins distinct GPUs only, so engines of one
// GPU first merge their keys on that GPU (k_max_u64) and one representative per GPU takes part in the all-reduce; with every
// engine on one GPU no RCCL call is made at all (and RCCL is not even loaded) -- which is also how the N > 1 control flow
// (partitions, per-engine buffers, merge, window restore) is tested on the one-GPU box.
//
// GPSACQ_MULTI_FORCE_RCCL=1 (read at gpsacq_multi_create) makes the communicator and the all-reduce happen whatever the
// device list: with one distinct GPU that is ncclCommInitAll(1) and a one-rank ncclAllReduce inside ncclGroupStart/End --
// the very calls an 8-GPU node makes, executed on the one-GPU box (tests/test_gpu_round4.py::test_rccl_single_rank_*).
//
// The caller's thread is not the critical path: each engine's share of a call -- staging copy of its part of the (pageable)
// capture into its own pinned buffer (block decomposition; the grid decomposition, where every engine needs the whole capture,
// stages it once in one portable pinned buffer), upload, search, key packing, download of its peaks into pinned memory -- is
// enqueued by its own host thread, so device i + 1 is searching while device i's copy is still running (round 3 did all of it on one
// thread through pageable hipMemcpyAsync, which blocks: every device waited for the ones before it).  gpsacq_multi_last_call_ms
// reports how long the enqueue phase of the last call took on the host next to the whole call.
//
// RCCL is loaded with dlopen at the first gpsacq_multi_create so that libgpsacq.so carries no link dependency on it
// (a process that already holds an RCCL -- PyTorch bundles one under the same SONAME -- keeps using that copy).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

#include "../../include/gpsacq.h"
#include "acq_launch.hpp"

using namespace acq;

namespace {
struct Rccl {
    void* so = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;             // published only when every symbol resolved
std::mutex g_rccl_lock;  // gpsacq_multi_create may be called from several threads
const char* load_rccl() {
    std::lock_guard<std::mutex> guard(g_rccl_lock);
    if (g_rccl.so) return nullptr;
    Rccl r;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        r.so = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (r.so) break;
    }
    if (!r.so) return "librccl.so.1 not found (dlopen)";
#define SYM(field, name)                                                 \
    r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.so, name));   \
    if (!r.field) {                                                      \
        dlclose(r.so);                                                   \
        return "RCCL symbol " name " missing";                           \
    }
    SYM(CommInitAll, "ncclCommInitAll")
    SYM(CommDestroy, "ncclCommDestroy")
    SYM(AllReduce, "ncclAllReduce")
    SYM(GroupStart, "ncclGroupStart")
    SYM(GroupEnd, "ncclGroupEnd")
    SYM(GetErrorString, "ncclGetErrorString")
#undef SYM
    g_rccl = r;
    return nullptr;
}
int failf(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    return acq::set_last_error(code, buf);
}
// the caller's current HIP device is the caller's business: put it back on every way out
struct DeviceGuard {
    int dev = -1;
    DeviceGuard() { if (hipGetDevice(&dev) != hipSuccess) dev = -1; }
    ~DeviceGuard() { if (dev >= 0) (void)hipSetDevice(dev); }
};
}  // namespace

struct gpsacq_multi {
    std::vector<gpsacq_engine*> eng;
    std::vector<int> dev;
    std::vector<int> rep;            // engine i's representative: the first engine on the same physical GPU
    std::vector<int> reps;           // the representatives (one engine per distinct GPU), engine 0 first
    std::vector<ncclComm_t> comm;    // per engine; non-NULL for representatives when more than one GPU takes part
    std::vector<hipEvent_t> ev;      // per engine: orders a partner's stream behind this engine's work
    std::vector<uint8_t*> d_bits;
    std::vector<size_t> bits_cap;
    std::vector<Task*> d_tasks;
    std::vector<Peak*> d_peaks;
    // per engine, each (task_cap + 32) entries: own keys, merged keys; own winner powers, merged winner powers
    std::vector<unsigned long long*> d_keys;
    std::vector<unsigned long long*> d_merged;
    std::vector<float*> d_pwr;
    std::vector<float*> d_pwr_merged;
    std::vector<size_t> task_cap;
    // pinned staging per engine: its share of the caller's capture on the way in, its peaks on the way out
    std::vector<uint8_t*> h_bits;
    std::vector<size_t> h_bits_cap;
    std::vector<Peak*> h_peaks;
    std::vector<size_t> h_peaks_cap;
    // grid decomposition: every engine uploads the SAME capture and task list -- staged once, in one pinned buffer that every
    // device can read (hipHostMallocPortable), not once per engine
    uint8_t* h_shared = nullptr;
    size_t h_shared_cap = 0;
    bool force_rccl = false;         // GPSACQ_MULTI_FORCE_RCCL=1: communicator + all-reduce even with one distinct GPU
    long rccl_allreduces = 0;        // ncclAllReduce calls issued so far (all ranks of a group count once)
    double last_enqueue_ms = 0, last_total_ms = 0;
    gpsacq_info info{};
};

#define HIPM(expr)                                                                                        \
    do {                                                                                                  \
        hipError_t e_ = (expr);                                                                           \
        if (e_ != hipSuccess) return failf(e_ == hipErrorOutOfMemory ? GPSACQ_ERR_NOMEM : GPSACQ_ERR_DEVICE, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

extern "C" void gpsacq_multi_destroy(gpsacq_multi* m) {
    if (!m) return;
    DeviceGuard guard;
    for (size_t i = 0; i < m->eng.size(); ++i) {
        if (!m->eng[i]) continue;  // creation stopped before this device
        (void)hipSetDevice(m->dev[i]);
        (void)gpsacq_synchronize(m->eng[i]);
        if (i < m->comm.size() && m->comm[i]) (void)g_rccl.CommDestroy(m->comm[i]);
        if (i < m->ev.size() && m->ev[i]) (void)hipEventDestroy(m->ev[i]);
        for (void* p : {(void*)m->d_bits[i], (void*)m->d_tasks[i], (void*)m->d_peaks[i], (void*)m->d_keys[i], (void*)m->d_merged[i],
                        (void*)m->d_pwr[i], (void*)m->d_pwr_merged[i]})
            if (p) (void)hipFree(p);
        if (m->h_bits[i]) (void)hipHostFree(m->h_bits[i]);
        if (m->h_peaks[i]) (void)hipHostFree(m->h_peaks[i]);
        gpsacq_destroy(m->eng[i]);
    }
    if (m->h_shared) (void)hipHostFree(m->h_shared);
    delete m;
}

extern "C" int gpsacq_multi_create(const gpsacq_params* params, const int32_t* devices, int n_devices, gpsacq_multi** out) {
    if (!params || !out || n_devices < 1 || n_devices > 64) return failf(GPSACQ_ERR_ARG, "gpsacq_multi_create: bad argument");
    *out = nullptr;
    DeviceGuard guard;
    gpsacq_multi* m = new gpsacq_multi();
    const size_t n = (size_t)n_devices;
    m->eng.assign(n, nullptr);
    m->dev.resize(n);
    m->rep.resize(n);
    m->comm.assign(n, nullptr);
    m->ev.assign(n, nullptr);
    m->d_bits.assign(n, nullptr);
    m->bits_cap.assign(n, 0);
    m->d_tasks.assign(n, nullptr);
    m->d_peaks.assign(n, nullptr);
    m->d_keys.assign(n, nullptr);
    m->d_merged.assign(n, nullptr);
    m->d_pwr.assign(n, nullptr);
    m->d_pwr_merged.assign(n, nullptr);
    m->task_cap.assign(n, 0);
    m->h_bits.assign(n, nullptr);
    m->h_bits_cap.assign(n, 0);
    m->h_peaks.assign(n, nullptr);
    m->h_peaks_cap.assign(n, 0);
    {
        const char* fr = getenv("GPSACQ_MULTI_FORCE_RCCL");
        m->force_rccl = fr && *fr && atoi(fr) != 0;
    }
    for (size_t i = 0; i < n; ++i) {
        m->dev[i] = devices ? devices[i] : (int)i;
        m->rep[i] = (int)i;
        for (size_t j = 0; j < i; ++j)
            if (m->dev[j] == m->dev[i]) {
                m->rep[i] = m->rep[j];
                break;
            }
        if (m->rep[i] == (int)i) m->reps.push_back((int)i);
    }
    for (size_t i = 0; i < n; ++i) {
        gpsacq_params p = *params;
        p.device = m->dev[i];
        if (int rc = gpsacq_create(&p, &m->eng[i])) {
            gpsacq_multi_destroy(m);
            return rc;
        }
        hipError_t he = hipEventCreateWithFlags(&m->ev[i], hipEventDisableTiming);
        if (he != hipSuccess) {
            int rc = failf(GPSACQ_ERR_DEVICE, "hipEventCreate: %s", hipGetErrorString(he));
            gpsacq_multi_destroy(m);
            return rc;
        }
    }
    if (m->reps.size() > 1 || m->force_rccl) {  // one communicator rank per distinct GPU
        if (const char* why = load_rccl()) {
            int rc = failf(GPSACQ_ERR_DEVICE, "RCCL unavailable: %s", why);
            gpsacq_multi_destroy(m);
            return rc;
        }
        std::vector<ncclComm_t> comms(m->reps.size(), nullptr);
        std::vector<int> devs;
        for (int r : m->reps) devs.push_back(m->dev[r]);
        ncclResult_t r = g_rccl.CommInitAll(comms.data(), (int)devs.size(), devs.data());
        if (r != ncclSuccess) {
            int rc = failf(GPSACQ_ERR_DEVICE, "ncclCommInitAll over %zu device(s): %s", devs.size(), g_rccl.GetErrorString(r));
            gpsacq_multi_destroy(m);
            return rc;
        }
        for (size_t k = 0; k < m->reps.size(); ++k) m->comm[m->reps[k]] = comms[k];
    }
    (void)gpsacq_get_info(m->eng[0], &m->info);
    *out = m;
    return GPSACQ_OK;
}

extern "C" int gpsacq_multi_set_doppler_step(gpsacq_multi* m, double step_hz) {
    if (!m) return failf(GPSACQ_ERR_ARG, "gpsacq_multi_set_doppler_step: null handle");
    DeviceGuard guard;
    for (gpsacq_engine* e : m->eng)
        if (int rc = gpsacq_set_doppler_step(e, step_hz)) return rc;
    (void)gpsacq_get_info(m->eng[0], &m->info);
    return GPSACQ_OK;
}

extern "C" int gpsacq_multi_get_info(const gpsacq_multi* m, gpsacq_info* info, int32_t* n_devices) {
    if (!m || !info) return failf(GPSACQ_ERR_ARG, "gpsacq_multi_get_info: null argument");
    *info = m->info;
    if (n_devices) *n_devices = (int32_t)m->eng.size();
    return GPSACQ_OK;
}

namespace {
// Per-device scratch, stream-ordered on that device's engine stream (no device-wide synchronisation in mid-pipeline):
// work already enqueued keeps the old buffer until it has run.
int grow_dev(gpsacq_multi* m, size_t i, size_t nbytes, size_t n_tasks) {
    hipStream_t st = (hipStream_t)gpsacq_stream(m->eng[i]);
    if (nbytes > m->bits_cap[i]) {
        if (m->d_bits[i]) HIPM(hipFreeAsync(m->d_bits[i], st));
        m->d_bits[i] = nullptr;
        m->bits_cap[i] = 0;
        HIPM(hipMallocAsync((void**)&m->d_bits[i], nbytes, st));
        m->bits_cap[i] = nbytes;
    }
    if (n_tasks > m->task_cap[i]) {
        // (a partner engine of the same GPU may still read the merge buffers of an earlier call: that call has been waited for)
        for (void** p : {(void**)&m->d_tasks[i], (void**)&m->d_peaks[i], (void**)&m->d_keys[i], (void**)&m->d_merged[i], (void**)&m->d_pwr[i],
                         (void**)&m->d_pwr_merged[i]}) {
            if (*p) HIPM(hipFreeAsync(*p, st));
            *p = nullptr;
        }
        m->task_cap[i] = 0;
        const size_t nk = n_tasks + GPSACQ_NUM_SATS;  // + the 32 per-PRN entries of the block decomposition
        HIPM(hipMallocAsync((void**)&m->d_tasks[i], n_tasks * sizeof(Task), st));
        HIPM(hipMallocAsync((void**)&m->d_peaks[i], n_tasks * sizeof(Peak), st));
        HIPM(hipMallocAsync((void**)&m->d_keys[i], nk * sizeof(unsigned long long), st));
        HIPM(hipMallocAsync((void**)&m->d_merged[i], nk * sizeof(unsigned long long), st));
        HIPM(hipMallocAsync((void**)&m->d_pwr[i], nk * sizeof(float), st));
        HIPM(hipMallocAsync((void**)&m->d_pwr_merged[i], nk * sizeof(float), st));
        m->task_cap[i] = n_tasks;
    }
    return GPSACQ_OK;
}
// wait for everything enqueued on every engine stream (also the way out of a failed call: nothing of it keeps running)
void drain(gpsacq_multi* m) {
    for (size_t i = 0; i < m->eng.size(); ++i) {
        (void)hipSetDevice(m->dev[i]);
        (void)hipStreamSynchronize((hipStream_t)gpsacq_stream(m->eng[i]));
    }
}
// pinned staging of engine i (host side of its uploads / downloads); calls are synchronous, so nothing is in flight when one grows
int grow_pinned(gpsacq_multi* m, size_t i, size_t nbytes, size_t n_peaks) {
    if (nbytes > m->h_bits_cap[i]) {
        if (m->h_bits[i]) HIPM(hipHostFree(m->h_bits[i]));
        m->h_bits[i] = nullptr;
        m->h_bits_cap[i] = 0;
        const size_t want = std::max(nbytes, (size_t)1 << 16);
        HIPM(hipHostMalloc((void**)&m->h_bits[i], want, hipHostMallocDefault));
        m->h_bits_cap[i] = want;
    }
    if (n_peaks > m->h_peaks_cap[i]) {
        if (m->h_peaks[i]) HIPM(hipHostFree(m->h_peaks[i]));
        m->h_peaks[i] = nullptr;
        m->h_peaks_cap[i] = 0;
        const size_t want = std::max(n_peaks, (size_t)1024);
        HIPM(hipHostMalloc((void**)&m->h_peaks[i], want * sizeof(Peak), hipHostMallocDefault));
        m->h_peaks_cap[i] = want;
    }
    return GPSACQ_OK;
}
// body(i) for every engine, each on its own host thread with that engine's device current (engine 0 on the calling thread):
// an engine is only ever touched by its own thread, gpsacq_last_error() is per thread, so the first failure's text is carried
// back to the caller's.  Threads that cannot be had (std::system_error) run inline -- slower, never wrong.
template <class F>
int for_each_engine(gpsacq_multi* m, F&& body) {
    const size_t n = m->eng.size();
    std::vector<int> rcs(n, GPSACQ_OK);
    std::vector<std::string> errs(n);
    auto run = [&](size_t i) noexcept {  // nothing may leave a worker thread (or the C ABI) as an exception
        try {
            int rc;
            const hipError_t he = hipSetDevice(m->dev[i]);
            if (he != hipSuccess) rc = failf(GPSACQ_ERR_DEVICE, "hipSetDevice(%d): %s", m->dev[i], hipGetErrorString(he));
            else rc = body(i);
            rcs[i] = rc;
            if (rc) errs[i] = gpsacq_last_error();
        } catch (...) {
            rcs[i] = GPSACQ_ERR_NOMEM;  // (errs[i] stays empty: its assignment may be what failed)
        }
    };
    std::vector<std::thread> workers;
    size_t threaded = 0;  // engines 1 .. threaded have a worker
    try {
        for (size_t i = 1; i < n; ++i) {
            workers.emplace_back(run, i);
            threaded = i;
        }
    } catch (...) {  // no thread to be had (std::system_error) or no memory for its handle: the rest runs inline below
    }
    run(0);
    for (size_t i = threaded + 1; i < n; ++i) run(i);
    for (std::thread& w : workers) w.join();
    for (size_t i = 0; i < n; ++i)
        if (rcs[i]) return acq::set_last_error(rcs[i], errs[i].empty() ? "out of memory on a worker thread" : errs[i].c_str());
    return GPSACQ_OK;
}
typedef std::chrono::steady_clock MClock;
double ms_since(MClock::time_point t0) { return std::chrono::duration<double, std::milli>(MClock::now() - t0).count(); }

// MAX-merge of one array per engine (64-bit keys, or float powers): engines of one GPU merge into their representative on
// the device, the representatives all-reduce over RCCL (the path's one exchange step between GPUs), and every engine gets
// the result back.  buf[i]: engine i's array (in: own values, out: merged), `count` entries.
template <class T>
int merge_max(gpsacq_multi* m, const std::vector<T*>& buf, size_t count) {
    const size_t n = m->eng.size();
    auto stream = [&](size_t i) { return (hipStream_t)gpsacq_stream(m->eng[i]); };
    auto launch_max = [&](T* dst, const T* src, hipStream_t st) {
        if constexpr (sizeof(T) == 8) launch_max_u64((unsigned long long*)dst, (const unsigned long long*)src, (int)count, st);
        else launch_max_f32((float*)dst, (const float*)src, (int)count, st);
    };
    // 1. partners -> representative, on the representative's stream, behind the partner's own work
    for (size_t j = 0; j < n; ++j) {
        if (m->rep[j] == (int)j) continue;
        const size_t r = (size_t)m->rep[j];
        HIPM(hipSetDevice(m->dev[j]));
        HIPM(hipEventRecord(m->ev[j], stream(j)));
        HIPM(hipStreamWaitEvent(stream(r), m->ev[j], 0));
        launch_max(buf[r], buf[j], stream(r));
        HIPM(hipGetLastError());
    }
    // 2. across GPUs (a one-rank group when GPSACQ_MULTI_FORCE_RCCL made a communicator for a single GPU)
    if (m->comm[(size_t)m->reps[0]]) {
        ++m->rccl_allreduces;
        ncclResult_t r = g_rccl.GroupStart();
        if (r != ncclSuccess) return failf(GPSACQ_ERR_DEVICE, "ncclGroupStart: %s", g_rccl.GetErrorString(r));
        ncclResult_t first_bad = ncclSuccess;
        int bad_dev = -1;
        for (int i : m->reps) {
            // every rank of the group is enqueued even after a failure: ending a group that some ranks never joined can hang
            r = g_rccl.AllReduce(buf[i], buf[i], count, sizeof(T) == 8 ? ncclUint64 : ncclFloat32, ncclMax, m->comm[i], stream((size_t)i));
            if (r != ncclSuccess && first_bad == ncclSuccess) {
                first_bad = r;
                bad_dev = m->dev[i];
            }
        }
        r = g_rccl.GroupEnd();
        if (first_bad != ncclSuccess) return failf(GPSACQ_ERR_DEVICE, "ncclAllReduce on device %d: %s", bad_dev, g_rccl.GetErrorString(first_bad));
        if (r != ncclSuccess) return failf(GPSACQ_ERR_DEVICE, "ncclGroupEnd: %s", g_rccl.GetErrorString(r));
    }
    // 3. representative -> partners
    for (int ri : m->reps) {
        bool has_partner = false;
        for (size_t j = 0; j < n; ++j) has_partner |= (m->rep[j] == ri && (int)j != ri);
        if (!has_partner) continue;
        HIPM(hipSetDevice(m->dev[ri]));
        HIPM(hipEventRecord(m->ev[ri], stream((size_t)ri)));
        for (size_t j = 0; j < n; ++j) {
            if (m->rep[j] != ri || (int)j == ri) continue;
            HIPM(hipStreamWaitEvent(stream(j), m->ev[ri], 0));
            HIPM(hipMemcpyAsync(buf[j], buf[ri], count * sizeof(T), hipMemcpyDeviceToDevice, stream(j)));
        }
    }
    return GPSACQ_OK;
}
// keys: own -> merged (copy, merge); then the winners' max_pwr the same way.  own_keys / own_pwr are per-engine arrays at `off`.
int merge_peaks(gpsacq_multi* m, size_t off, size_t count) {
    const size_t n = m->eng.size();
    std::vector<unsigned long long*> mk(n);
    std::vector<float*> mp(n);
    for (size_t i = 0; i < n; ++i) {
        HIPM(hipSetDevice(m->dev[i]));
        hipStream_t st = (hipStream_t)gpsacq_stream(m->eng[i]);
        mk[i] = m->d_merged[i] + off;
        mp[i] = m->d_pwr_merged[i] + off;
        HIPM(hipMemcpyAsync(mk[i], m->d_keys[i] + off, count * sizeof(unsigned long long), hipMemcpyDeviceToDevice, st));
    }
    if (int rc = merge_max(m, mk, count)) return rc;
    for (size_t i = 0; i < n; ++i) {
        HIPM(hipSetDevice(m->dev[i]));
        launch_winner_pwr(m->d_keys[i] + off, mk[i], m->d_pwr[i] + off, mp[i], (int)count, (hipStream_t)gpsacq_stream(m->eng[i]));
        HIPM(hipGetLastError());
    }
    return merge_max(m, mp, count);
}
void unpack_key(unsigned long long k, float pwr, int kmax, gpsacq_peak* p) {
    const uint32_t sb = (uint32_t)(k >> 32);
    float snr;
    memcpy(&snr, &sb, sizeof snr);
    p->snr = snr;
    p->lo_shift = k ? (int32_t)(0xFFFF - ((k >> 16) & 0xFFFF)) - kmax : 0;
    p->ca_shift = (int32_t)(k & 0xFFFF);
    p->max_pwr = pwr;  // not carried by the key: merged separately (the winner's value)
}
}  // namespace

static int search_grid_impl(gpsacq_multi* m, const uint8_t* bits, size_t n_blocks, size_t stride, const gpsacq_task* tasks,
                            size_t n_tasks, gpsacq_peak* peaks) {
    const MClock::time_point t_call = MClock::now();
    const size_t n = m->eng.size();
    const size_t nbytes = (n_blocks - 1) * stride + (stride < (size_t)GPSACQ_BLOCK_BYTES ? stride : (size_t)GPSACQ_BLOCK_BYTES);
    const int total = m->info.num_doppler_total, first = m->info.first_doppler_total, kmax = -first;
    // capture and task list (tasks behind the capture, 16-byte aligned) staged ONCE for all engines: one host copy and one
    // pinned buffer whatever the number of devices (calls are synchronous: nothing of an earlier call still reads it)
    const size_t task_off = (nbytes + 15) & ~(size_t)15, shared_bytes = task_off + n_tasks * sizeof(Task);
    if (shared_bytes > m->h_shared_cap) {
        if (m->h_shared) HIPM(hipHostFree(m->h_shared));
        m->h_shared = nullptr;
        m->h_shared_cap = 0;
        const size_t want = std::max(shared_bytes, (size_t)1 << 16);
        HIPM(hipHostMalloc((void**)&m->h_shared, want, hipHostMallocPortable));
        m->h_shared_cap = want;
    }
    memcpy(m->h_shared, bits, nbytes);
    memcpy(m->h_shared + task_off, tasks, n_tasks * sizeof(Task));
    const int rc_enq = for_each_engine(m, [&](size_t i) -> int {
        // contiguous, balanced slab of the grid for device i (possibly empty when there are more devices than points)
        const int base = total / (int)n, rem = total % (int)n;
        const int cnt = base + ((int)i < rem ? 1 : 0), off = (int)i * base + ((int)i < rem ? (int)i : rem);
        hipStream_t st = (hipStream_t)gpsacq_stream(m->eng[i]);
        if (int rc = grow_dev(m, i, nbytes, n_tasks)) return rc;
        if (cnt > 0) {
            HIPM(hipMemcpyAsync(m->d_bits[i], m->h_shared, nbytes, hipMemcpyHostToDevice, st));
            HIPM(hipMemcpyAsync(m->d_tasks[i], m->h_shared + task_off, n_tasks * sizeof(Task), hipMemcpyHostToDevice, st));
            if (int rc = gpsacq_set_doppler_window(m->eng[i], first + off, cnt)) return rc;
            if (int rc = gpsacq_search_device(m->eng[i], m->d_bits[i], n_blocks, stride, m->d_tasks[i], n_tasks, nullptr, m->d_peaks[i], 0)) return rc;
            launch_pack_keys(m->d_peaks[i], m->d_keys[i], (int)n_tasks, kmax, st);
            launch_peak_pwr(m->d_peaks[i], m->d_pwr[i], (int)n_tasks, st);
            HIPM(hipGetLastError());
        } else {
            HIPM(hipMemsetAsync(m->d_keys[i], 0, n_tasks * sizeof(unsigned long long), st));  // key 0 = "nothing found": neutral for MAX
            HIPM(hipMemsetAsync(m->d_pwr[i], 0, n_tasks * sizeof(float), st));
        }
        return GPSACQ_OK;
    });
    if (rc_enq) return rc_enq;
    if (int rc = merge_peaks(m, 0, n_tasks)) return rc;
    m->last_enqueue_ms = ms_since(t_call);
    std::vector<unsigned long long> keys(n_tasks);
    std::vector<float> pwr(n_tasks);
    HIPM(hipSetDevice(m->dev[0]));
    HIPM(hipMemcpyAsync(keys.data(), m->d_merged[0], n_tasks * sizeof(unsigned long long), hipMemcpyDeviceToHost, (hipStream_t)gpsacq_stream(m->eng[0])));
    HIPM(hipMemcpyAsync(pwr.data(), m->d_pwr_merged[0], n_tasks * sizeof(float), hipMemcpyDeviceToHost, (hipStream_t)gpsacq_stream(m->eng[0])));
    for (size_t i = 0; i < n; ++i) {
        HIPM(hipSetDevice(m->dev[i]));
        HIPM(hipStreamSynchronize((hipStream_t)gpsacq_stream(m->eng[i])));
    }
    for (size_t t = 0; t < n_tasks; ++t) unpack_key(keys[t], pwr[t], kmax, &peaks[t]);
    m->last_total_ms = ms_since(t_call);
    return GPSACQ_OK;
}

extern "C" int gpsacq_multi_search_grid(gpsacq_multi* m, const uint8_t* bits, size_t n_blocks, size_t stride, const gpsacq_task* tasks,
                                        size_t n_tasks, gpsacq_peak* peaks) {
    if (!m || !bits || !tasks || !peaks || n_blocks == 0 || n_tasks == 0 || stride < 5000) return failf(GPSACQ_ERR_ARG, "gpsacq_multi_search_grid: bad argument");
    for (size_t t = 0; t < n_tasks; ++t)
        if (tasks[t].block < 0 || (size_t)tasks[t].block >= n_blocks || tasks[t].prn < 0 || tasks[t].prn >= GPSACQ_NUM_SATS)
            return failf(GPSACQ_ERR_ARG, "task %zu = (block %d, prn %d) out of range", t, tasks[t].block, tasks[t].prn);
    DeviceGuard guard;
    // the slabs are set through each engine's Doppler window: remember the windows and put them back whatever happens
    std::vector<gpsacq_info> before(m->eng.size());
    for (size_t i = 0; i < m->eng.size(); ++i) (void)gpsacq_get_info(m->eng[i], &before[i]);
    const int rc = search_grid_impl(m, bits, n_blocks, stride, tasks, n_tasks, peaks);
    std::string err = rc ? gpsacq_last_error() : "";
    if (rc) drain(m);  // nothing of a failed call keeps running on any device
    for (size_t i = 0; i < m->eng.size(); ++i) (void)gpsacq_set_doppler_window(m->eng[i], before[i].first_doppler, before[i].num_doppler);
    if (rc) return acq::set_last_error(rc, err.c_str());
    return GPSACQ_OK;
}

// ---- the block decomposition: whole runs of the reference schedule split over the devices -------------------------------
static int search_blocks_impl(gpsacq_multi* m, const uint8_t* bits, size_t n_runs, size_t stride, gpsacq_peak* peaks, gpsacq_peak* best) {
    const MClock::time_point t_call = MClock::now();
    const size_t n = m->eng.size();
    const int first = m->info.first_doppler_total, total = m->info.num_doppler_total, kmax = -first;
    size_t off = 0;  // the 32 per-PRN entries live behind the per-task entries: the same offset on every engine
    {
        const size_t base = n_runs / n, rem = n_runs % n;
        off = std::max((base + (rem ? 1 : 0)) * GPSACQ_NUM_SATS, (size_t)1);
        for (size_t i = 0; i < n; ++i) off = std::max(off, m->task_cap[i]);
    }
    // contiguous, balanced range of whole runs for device i; a run starts at PRN index 0, so the reference schedule
    // (block t <-> PRN t % 32) holds inside every range
    auto share = [&](size_t i, size_t& first_run, size_t& nblk) {
        const size_t base = n_runs / n, rem = n_runs % n;
        const size_t cnt = base + (i < rem ? 1 : 0);
        first_run = i * base + (i < rem ? i : rem);
        nblk = cnt * GPSACQ_NUM_SATS;
    };
    const int rc_enq = for_each_engine(m, [&](size_t i) -> int {
        size_t first_run, nblk;
        share(i, first_run, nblk);
        hipStream_t st = (hipStream_t)gpsacq_stream(m->eng[i]);
        const size_t nbytes = nblk ? (nblk - 1) * stride + GPSACQ_BLOCK_BYTES : 0;
        if (int rc = grow_dev(m, i, std::max(nbytes, (size_t)1), off)) return rc;
        if (nblk > 0) {
            if (int rc = grow_pinned(m, i, nbytes, peaks ? nblk : 0)) return rc;
            memcpy(m->h_bits[i], bits + first_run * GPSACQ_NUM_SATS * stride, nbytes);  // this engine's share, by this engine's thread
            HIPM(hipMemcpyAsync(m->d_bits[i], m->h_bits[i], nbytes, hipMemcpyHostToDevice, st));
            if (int rc = gpsacq_set_doppler_window(m->eng[i], first, total)) return rc;  // every device scans the whole grid
            if (int rc = gpsacq_search_device(m->eng[i], m->d_bits[i], nblk, stride, nullptr, nblk, nullptr, m->d_peaks[i], 0)) return rc;
            launch_pack_keys(m->d_peaks[i], m->d_keys[i], (int)nblk, kmax, st);
            launch_prn_best(m->d_keys[i], m->d_peaks[i], (int)nblk, m->d_keys[i] + off, m->d_pwr[i] + off, st);
            HIPM(hipGetLastError());
            if (peaks) HIPM(hipMemcpyAsync(m->h_peaks[i], m->d_peaks[i], nblk * sizeof(Peak), hipMemcpyDeviceToHost, st));  // pinned: does not wait
        } else {
            HIPM(hipMemsetAsync(m->d_keys[i] + off, 0, GPSACQ_NUM_SATS * sizeof(unsigned long long), st));  // neutral for MAX
            HIPM(hipMemsetAsync(m->d_pwr[i] + off, 0, GPSACQ_NUM_SATS * sizeof(float), st));
        }
        return GPSACQ_OK;
    });
    if (rc_enq) return rc_enq;
    if (int rc = merge_peaks(m, off, GPSACQ_NUM_SATS)) return rc;
    m->last_enqueue_ms = ms_since(t_call);  // everything of this call is enqueued on every device
    unsigned long long keys[GPSACQ_NUM_SATS];
    float pwr[GPSACQ_NUM_SATS];
    HIPM(hipSetDevice(m->dev[0]));
    HIPM(hipMemcpyAsync(keys, m->d_merged[0] + off, sizeof keys, hipMemcpyDeviceToHost, (hipStream_t)gpsacq_stream(m->eng[0])));
    HIPM(hipMemcpyAsync(pwr, m->d_pwr_merged[0] + off, sizeof pwr, hipMemcpyDeviceToHost, (hipStream_t)gpsacq_stream(m->eng[0])));
    for (size_t i = 0; i < n; ++i) {
        HIPM(hipSetDevice(m->dev[i]));
        HIPM(hipStreamSynchronize((hipStream_t)gpsacq_stream(m->eng[i])));
    }
    if (peaks)
        for (size_t i = 0; i < n; ++i) {
            size_t first_run, nblk;
            share(i, first_run, nblk);
            if (nblk) memcpy(peaks + first_run * GPSACQ_NUM_SATS, m->h_peaks[i], nblk * sizeof(Peak));
        }
    if (best)
        for (int sv = 0; sv < GPSACQ_NUM_SATS; ++sv) unpack_key(keys[sv], pwr[sv], kmax, &best[sv]);
    m->last_total_ms = ms_since(t_call);
    return GPSACQ_OK;
}

extern "C" int gpsacq_multi_search_blocks(gpsacq_multi* m, const uint8_t* bits, size_t n_runs, size_t stride, gpsacq_peak* peaks, gpsacq_peak* best) {
    if (!m || !bits || n_runs == 0 || stride < (size_t)GPSACQ_BLOCK_BYTES || (!peaks && !best))
        return failf(GPSACQ_ERR_ARG, "gpsacq_multi_search_blocks: bad argument (whole runs of 32 blocks, stride >= 5120)");
    DeviceGuard guard;
    std::vector<gpsacq_info> before(m->eng.size());
    for (size_t i = 0; i < m->eng.size(); ++i) (void)gpsacq_get_info(m->eng[i], &before[i]);
    const int rc = search_blocks_impl(m, bits, n_runs, stride, peaks, best);
    std::string err = rc ? gpsacq_last_error() : "";
    if (rc) drain(m);
    for (size_t i = 0; i < m->eng.size(); ++i) (void)gpsacq_set_doppler_window(m->eng[i], before[i].first_doppler, before[i].num_doppler);
    if (rc) return acq::set_last_error(rc, err.c_str());
    return GPSACQ_OK;
}

extern "C" int gpsacq_multi_last_call_ms(const gpsacq_multi* m, double* enqueue_ms, double* total_ms, int64_t* rccl_allreduces) {
    if (!m) return failf(GPSACQ_ERR_ARG, "gpsacq_multi_last_call_ms: null handle");
    if (enqueue_ms) *enqueue_ms = m->last_enqueue_ms;
    if (total_ms) *total_ms = m->last_total_ms;
    if (rccl_allreduces) *rccl_allreduces = (int64_t)m->rccl_allreduces;
    return GPSACQ_OK;
}
