/*
 * gpsacq.h -- C ABI of the MI355X (gfx950) GPS L1 C/A acquisition engine.
 *
 * This is the drop-in boundary for the hot path of the reference's offline search stage
 * (JiaoXianjun/GNSS-GPS-SDR, c/search_offline.cpp).  Plain C: opaque handle, plain pointers
 * and sizes, int status codes (0 = ok), caller-allocated outputs, no exceptions cross it.
 * The reference has no FFI (it is one C++ translation unit); what a maintainer would bind is
 * the set of functions declared in c/gps_offline.h:87-91, and each entry point below names
 * the reference function whose work it performs:
 *
 *   gpsacq_create            SearchInit()            c/search_offline.cpp:74-110
 *                            (+ the globals FC, FS, max_fo of c/gps_offline.h:23-25)
 *   gpsacq_destroy           SearchFree()            c/search_offline.cpp:114-117
 *   gpsacq_search            Sample() + Correlate()  c/search_offline.cpp:121-165, 169-201
 *                            for a batch of 5120-byte blocks (the body of the
 *                            SearchTask() loop, :239-246)
 *   gpsacq_search_device     same, capture already resident in HBM
 *   gpsacq_peak_keys_device  Correlate()'s ordering of two results, :196-198 (strict '>' over ascending dop), as 64-bit keys whose
 *                            integer MAX merges the per-PRN best of several GPUs / ranks
 *   gpsacq_pipe_*            same, batches in flight: the SearchTask() loop's fread of batch k+1 overlaps the search of batch k
 *   gpsacq_search_iq8        same on an 8-bit IQ capture: what proc_rtl_bin_for_gps.m / proc_hackrf_bin_for_gps.m + gps_test
 *                            do in two steps through a 1-bit file, fused into the forward transform
 *   gpsacq_set_doppler_step  the Doppler grid of Correlate()'s loop, :176,182 (finer or coarser than fs/40000)
 *   gpsacq_set_cell_handout  how the iterations of Correlate()'s `for dop` loop (:176) and of SearchTask()'s `for sv` loop (:239) reach the
 *                            compute units: drawn at run time by persistent workgroups (default) or one workgroup per cell; same results
 *   gpsacq_multi_search_grid the same Correlate() grid cut over several GPUs, peaks merged by one RCCL all-reduce
 *   gpsacq_multi_search_blocks  the SearchTask() run loop (:237-262) cut over several GPUs by whole runs, per-PRN best
 *                            peak merged by one RCCL all-reduce
 *   gpsacq_search_code       SearchCode()            c/search_offline.cpp:205-209
 *   gpsacq_iq8_to_bits       the MATLAB pre-processing that produces gps_test's input from an 8-bit IQ
 *                            capture: proc_rtl_bin_for_gps.m:12-26,31-47, proc_hackrf_bin_for_gps.m:7-19
 *   gpsacq_generate          the role of gps_sig_gen.m (synthetic 1-bit capture; here noise + any PRN set with Doppler)
 *   gpsacq_generate_range    any byte window of that stream (a function of the absolute sample index): what lets every rank of a
 *                            multi-GPU job make its own blocks of the ONE capture all world sizes search
 *   gpsacq_generate_sig      gps_sig_gen.m:8-41 itself (one PRN, navigation bits, raised-cosine BPSK at fs/4), bit-exact
 *   gpsacq_generate_sig_tx   gps_sig_gen.m:21-30, the script's int8 complex-baseband file for HackRF replay
 *   gpsacq_handoff           CHANNEL::Start()'s NCO set-up from a search hit, c/channel.cpp:134-163
 *                            (the first consumer of the search result in the online receiver)
 *   gpsacq_sample_spectrum   Sample()'s fwd_buf      c/search_offline.cpp:161 (parity probe)
 *   gpsacq_code_spectrum     SearchInit()'s code[sv] c/search_offline.cpp:105-106 (parity probe)
 *
 * The C++-linkage mirror of the reference API itself (SearchInit/SearchTask/... consuming
 * the caller's FC/FS/max_fo globals) is include/gps_search.h, implemented on top of this ABI.
 */
#ifndef GPSACQ_H
#define GPSACQ_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* the library is built with -fvisibility=hidden: exactly the functions declared here are exported */
#define GPSACQ_API __attribute__((visibility("default")))

#define GPSACQ_FFT_LEN 40000      /* FFT_LEN, c/gps_offline.h:15 (compile-time in the reference) */
#define GPSACQ_NUM_SATS 32        /* NUM_SATS, c/gps_offline.h:16 */
#define GPSACQ_BLOCK_BYTES 5120   /* bytes consumed per Sample(): 10 packets x 512 B */

/* status codes */
#define GPSACQ_OK 0
#define GPSACQ_ERR_ARG 1          /* bad argument */
#define GPSACQ_ERR_DEVICE 2       /* no usable gfx950 device / HIP runtime error */
#define GPSACQ_ERR_UNSUPPORTED 3  /* parameter outside what the kernels cover: max_fo >= fs/2, ref_quirks + non-coherent */
#define GPSACQ_ERR_NOMEM 4

/* One engine = one device + one HIP stream + its scratch.  An engine is NOT thread-safe (the
 * reference's search stage is a single-instance, non-reentrant module as well, c/search_offline.cpp:55-64):
 * use one engine per host thread / per GPU.  gpsacq_last_error() is per thread. */
typedef struct gpsacq_engine gpsacq_engine;

typedef struct {
    double fc;          /* carrier (2nd IF) frequency in Hz      -- global FC  */
    double fs;          /* sampling rate in Hz                   -- global FS  */
    double max_fo;      /* Doppler search half-range in Hz       -- global max_fo */
    int32_t device;     /* HIP device ordinal */
    int32_t ref_quirks; /* 1: reproduce the reference's fwd_buf overrun, which replaces
                           code[0][0..959] by the block's samples 40000..40959 (PRN index 0
                           only; g++ BSS layout).  0: well-defined behaviour. */
} gpsacq_params;

/* one (block, PRN, Doppler bin): what Correlate()'s inner loop computes (:178-196) */
typedef struct {
    float max_pwr;   /* largest |IFFT|^2 over the first FS/1000 lags */
    int32_t max_i;   /* its lag (first one on ties) */
    float tot_pwr;   /* sum of |IFFT|^2 over those lags */
    float snr;       /* max_pwr / (tot_pwr / lags) */
} gpsacq_cell;

/* one (block, PRN): Correlate()'s return value and out-parameters (:196-200) */
typedef struct {
    float snr;         /* best SNR over the Doppler bins (0 if none > 0) */
    int32_t lo_shift;  /* Doppler bin of the best SNR, in units of fs/40000 Hz */
    int32_t ca_shift;  /* code phase (lag, samples) of the best SNR */
    float max_pwr;     /* max_pwr of that cell */
} gpsacq_peak;

/* one search task: block index into the capture, PRN index 0..31 */
typedef struct {
    int32_t block;
    int32_t prn;
} gpsacq_task;

typedef struct {
    int32_t fft_len;     /* 40000 */
    int32_t dmax;        /* Doppler bins searched are -dmax..+dmax (:176) */
    int32_t num_doppler; /* bins searched per task: 2*dmax+1 unless a window is set */
    int32_t first_doppler; /* first bin searched: -dmax unless a window is set */
    int32_t num_lags;    /* lags scanned per cell (:190) */
    int32_t acc_columns; /* kernel instance in use */
    int32_t device;
    int32_t compute_units;
    char device_name[64];
    /* Doppler grid (gpsacq_set_doppler_step); reference grid: sub = stride = 1, step = fs/40000 */
    int32_t doppler_sub;         /* sub-bin offsets per FFT bin (step = bin / sub) */
    int32_t doppler_stride;      /* or whole bins per grid point (step = bin * stride) */
    int32_t num_doppler_total;   /* grid points of the full +-max_fo range */
    int32_t first_doppler_total; /* index of the lowest one (= -(num_doppler_total - 1) / 2) */
    double doppler_step_hz;      /* lo_shift, first_doppler, Doppler windows are in units of this step */
} gpsacq_info;

/* per-stage device time of the last gpsacq_search* call, milliseconds (HIP events on the
 * engine's stream) and launch counts -- used by bench.py for the roofline line */
typedef struct {
    float ms_total;
    float ms_sample;     /* unpack + mix + forward FFT (both kernels) */
    float ms_correlate;  /* correlate kernel launches only */
    float ms_peaks;
    int32_t correlate_launches;
    int64_t cells;
} gpsacq_timing;

/* SearchInit(): builds the 32 code spectra, LO tables and twiddles on the device. */
GPSACQ_API int gpsacq_create(const gpsacq_params* params, gpsacq_engine** out);
/* SearchFree() */
GPSACQ_API void gpsacq_destroy(gpsacq_engine* e);
/* message of the last failing call on this thread ("" if none) */
GPSACQ_API const char* gpsacq_last_error(void);
GPSACQ_API int gpsacq_get_info(const gpsacq_engine* e, gpsacq_info* info);

/*
 * Search a batch.  `bits`: capture bytes, 1-bit real IF samples packed LSB first; block b
 * starts at bits + b*stride and 5120 bytes of it are read (5000 transformed; the last 120
 * only matter with ref_quirks).  `tasks`: n_tasks (block, prn) pairs, or NULL for the
 * reference's schedule: task t = (block t, prn t % 32) with n_tasks == n_blocks
 * (SearchTask(), :239-246).  Outputs (either may be NULL): cells[n_tasks][num_doppler] in
 * ascending Doppler-bin order, peaks[n_tasks].  Host pointers; returns after completion.
 */
GPSACQ_API int gpsacq_search(gpsacq_engine* e, const uint8_t* bits, size_t n_blocks, size_t stride,
                  const gpsacq_task* tasks, size_t n_tasks, gpsacq_cell* cells, gpsacq_peak* peaks);

/*
 * Same with every buffer already in this device's memory (bits, tasks, cells, peaks are
 * device pointers; tasks may be NULL as above; cells may be NULL).  Work is enqueued on the
 * engine's stream; `sync` != 0 waits for completion.  A device task list is not read by the host:
 * a task whose block or prn is out of range is skipped by the kernel and reported as cells with
 * max_i = -1, snr = 0 (peak: snr = 0) instead of GPSACQ_ERR_ARG.
 */
GPSACQ_API int gpsacq_search_device(gpsacq_engine* e, const void* d_bits, size_t n_blocks, size_t stride,
                         const void* d_tasks, size_t n_tasks, void* d_cells, void* d_peaks, int sync);
/*
 * Pipelined host-buffer searches on the reference schedule (task t = block t against PRN t % 32, the body of SearchTask()'s
 * loop :237-262).  A slot owns a pinned host staging buffer, a device copy and a pinned peak array:
 *   buf = gpsacq_pipe_buffer(e, slot, nbytes)      pinned buffer of >= nbytes (fread the batch straight into it); NULL on error.  A call
 *                                                  that asks for more than the slot holds re-allocates it: pointers from earlier calls
 *                                                  for that slot are then invalid (ask once for the largest batch)
 *   gpsacq_pipe_submit(e, slot, n_blocks, stride, iq)   upload on a second stream + search, returns at once (iq == NULL: 1-bit
 *                                                  blocks `stride` bytes apart; else 8-bit IQ, see gpsacq_search_iq8)
 *   gpsacq_pipe_collect(e, slot, &peaks, &n)       waits for THAT slot's search; peaks stay valid until its next submit
 * Searches run in submit order.  With 2-3 slots the caller's read of batch k+1 and report of batch k-1 overlap batch k.
 */
GPSACQ_API int gpsacq_reserve(gpsacq_engine* e, size_t n_blocks); /* scratch for batches of up to n_blocks blocks, once */
#define GPSACQ_PIPE_SLOTS 3
struct gpsacq_iq8_input;
GPSACQ_API uint8_t* gpsacq_pipe_buffer(gpsacq_engine* e, int slot, size_t nbytes);
GPSACQ_API int gpsacq_pipe_submit(gpsacq_engine* e, int slot, size_t n_blocks, size_t stride, const struct gpsacq_iq8_input* iq);
GPSACQ_API int gpsacq_pipe_collect(gpsacq_engine* e, int slot, const gpsacq_peak** peaks, size_t* n_peaks);

/*
 * Restrict the search to Doppler bins first_bin .. first_bin+n_bins-1 (within -dmax..+dmax).
 * Used to shard one block's PRN x Doppler grid over several GPUs (no reference equivalent: the
 * reference always scans the full range, :176).  cells rows then hold n_bins entries and
 * lo_shift stays an absolute bin number.
 */
GPSACQ_API int gpsacq_set_doppler_window(gpsacq_engine* e, int first_bin, int n_bins);
/*
 * How the correlate kernel's cells reach the compute units (no reference equivalent: Correlate(), c/search_offline.cpp:169-201, is a
 * serial loop).  on != 0 (the default; GPSACQ_CORR_PERSIST=0 in the environment at gpsacq_create turns it off): three resident
 * workgroups per CU draw (task, Doppler point) tickets at run time -- every XCD of the package takes work at its own pace, a task's
 * points (or a ~128-point chunk of a fine grid's task) stay on one XCD.  on == 0: one workgroup per cell, block g on XCD g % 8, as up
 * to round 5.  The cells are the same bit for bit either way (the same arithmetic per cell); only the time differs (3-5 % by box).
 * Applies to the instances that run three workgroups per CU (every coherent search up to fs = 8.25 MHz, and the 12-column
 * non-coherent one without re-alignment); the others ignore it.
 */
GPSACQ_API int gpsacq_set_cell_handout(gpsacq_engine* e, int on);
/*
 * Doppler grid step (extension; the reference's grid is whole FFT bins of fs/40000 Hz, c/search_offline.cpp:176,182,
 * and its front end ignores argv[4]).  step_hz <= 0 or within (bin, 2 bin): the reference grid.  step_hz < bin: the
 * grid is refined to bin / R, R = ceil(bin / step_hz) (the finest grid not coarser than asked): each block is
 * transformed R times, spectrum r being that of the block multiplied by exp(-2 pi i (r/R) n / 40000) -- a carrier
 * offset of r/R of a bin, exact, folded into the forward transform's twiddles -- and grid point k = d R + r pairs
 * spectrum r with the code spectrum shifted by d whole bins.  step_hz >= 2 bin: every S-th bin, S = floor(step_hz /
 * bin).  The range is -K..+K with K = trunc(max_fo / step) like :176.  Afterwards num_doppler / first_doppler /
 * lo_shift / Doppler windows count grid points (Hz = index * gpsacq_info.doppler_step_hz), cells rows hold one
 * record per grid point in ascending frequency, and the Doppler window is reset to the full range.
 * Costs R forward transforms and R x 320 KB of spectra per block; the correlate work per grid point is unchanged.
 */
#define GPSACQ_MAX_DOPPLER_SUB 16
GPSACQ_API int gpsacq_set_doppler_step(gpsacq_engine* e, double step_hz);
/*
 * Non-coherent accumulation (extension; the reference scans one coherent block per cell,
 * c/search_offline.cpp:176-199): every task then sums |IFFT|^2 per lag over n_acc block spectra
 * task.block + k*block_step (k < n_acc) before the peak scan; cells/peaks describe the summed power.
 * Lags only line up if consecutive accumulated blocks start a whole number of C/A periods apart:
 * lay the capture out with stride = gpsacq_aligned_stride(e) bytes (smallest multiple of
 * FS/8000 bytes >= 5120; -1 if FS/1000 is not a multiple of 8).  n_acc = 1 restores the
 * reference behaviour.  With tasks == NULL the schedule is task t = (block t, prn t % 32) for
 * t < n_tasks <= n_blocks - (n_acc-1)*block_step.
 */
GPSACQ_API int gpsacq_set_noncoherent(gpsacq_engine* e, int n_acc, int block_step);
/*
 * Code-creep compensation for the non-coherent mode (off by default).  A carrier offset f -- true
 * Doppler, or the crystal error of a single-oscillator front end such as the rtl-sdr, which is what
 * the +-100 kHz search range is for -- comes with a code-rate offset f/L1, so the correlation peak of
 * accumulated block k sits k * T * f / L1 samples later than block 0's (T = samples between
 * accumulated blocks = block_step * stride * 8): 1.07 samples per block at 40 kHz and fs = 2.8 MHz.
 * With compensation on, block k's powers are moved back by round(k * c * bin) whole samples (modulo
 * the fs/1000 lags) before they are summed, c = T * (fs/40000) / 1575.42e6 in float, bin = the cell's
 * Doppler bin; ca_shift then refers to block 0.  At fs > 10 MHz (more than 10000 lags, searched in
 * several passes of 40 columns) the per-lag sums are kept in device memory -- 4 * fs/1000 bytes per cell, the batch's tasks taken
 * in chunks of at most 1 GB of sums.
 */
GPSACQ_API int gpsacq_set_creep_compensation(gpsacq_engine* e, int on);
/*
 * Block alignment for the non-coherent mode (off by default; SURVEY.md section 8d, configs[3]: "per-block lag re-alignment").
 * Accumulated blocks whose starts are T = block_step * stride * 8 samples apart (stride / 2 for an 8-bit IQ capture) see the
 * code T mod (fs/1000) samples further on each time; with alignment on, block k's powers are moved back by k * (T mod S)
 * samples (modulo the S = fs/1000 lags) before they are summed, so any stride -- the file's own 5120-byte blocks, an IQ
 * capture's 81920 -- accumulates in phase and ca_shift refers to block 0.  gpsacq_aligned_stride() remains the layout that needs
 * no re-alignment (and lets the sums stay in registers).  Needs fs to be a whole number of samples per code period (fs = 1000 S).
 */
GPSACQ_API int gpsacq_set_block_alignment(gpsacq_engine* e, int on);
GPSACQ_API int gpsacq_aligned_stride(const gpsacq_engine* e);
/*
 * The merge keys of the multi-GPU reduction (the reference is one thread, c/search_offline.cpp has no
 * counterpart), made from the peaks of a gpsacq_*_device search on the engine's stream, behind that search, without a host wait:
 *   key = snr bits << 32 | (0xFFFF - (lo_shift + K)) << 16 | ca_shift,  K = -gpsacq_info.first_doppler_total
 * so that integer MAX over keys = "higher SNR, ties to the LOWER Doppler point" -- the strict '>' scan of :196-198.  (SNR >= 0:
 * the keys order the same as signed or unsigned 64-bit integers.)
 *   per_prn != 0: d_keys[32] = the best key of every PRN over the peaks t with t % 32 == PRN (the reference schedule, task t
 *                 <-> PRN t % 32, :239-246): what ONE all-reduce(MAX) of 256 bytes merges between the ranks of the block
 *                 decomposition.  n_peaks == 0 (a rank without work) writes 32 zero keys, neutral for MAX.
 *   per_prn == 0: d_keys[n_peaks], one key per peak (Doppler-slab decomposition: every rank holds every task).
 * Device pointers; d_peaks as written by gpsacq_search_device / gpsacq_search_iq8_device on this engine.
 */
GPSACQ_API int gpsacq_peak_keys_device(gpsacq_engine* e, const void* d_peaks, size_t n_peaks, int per_prn, void* d_keys, int sync);
GPSACQ_API int gpsacq_synchronize(gpsacq_engine* e);
/* stage times of the most recent search (waits for it to finish) ... */
GPSACQ_API int gpsacq_last_timing(const gpsacq_engine* e, gpsacq_timing* t);
/* ... and of the search n_back calls earlier (0 = last; the 8 most recent are kept): lets a caller
 * that enqueues searches with sync = 0 read a finished search's times while the next one runs */
GPSACQ_API int gpsacq_timing_ago(const gpsacq_engine* e, int n_back, gpsacq_timing* t);
/* Measurement aid next to gpsacq_last_timing (no reference counterpart): enqueues, on the engine's stream, a small kernel that writes
 * the shader-cycle counter (s_memtime) of every compute unit to d_stamp[XCC id << 6 | SE id << 4 | CU id] (device memory,
 * GPSACQ_STAMP_SLOTS x 8 bytes; zero it first: a slot no wave reached keeps its zero).  The counter is per compute unit -- its own
 * offset, and it stands still while the CU is clock-gated -- so only differences of the same slot mean anything.  Two calls around
 * a stretch of work that keeps the CUs busy, divided by the time between them, give the average shader clock every CU held over
 * that stretch: what bench.py's box-independent roofline (cycles per cell) is made of. */
#define GPSACQ_STAMP_SLOTS 512
GPSACQ_API int gpsacq_cycle_stamp_device(gpsacq_engine* e, void* d_stamp, int sync);
/* the engine's HIP stream (a hipStream_t) for callers that order their own device work after an
 * asynchronous gpsacq_*_device call with an event instead of a host wait; NULL for a NULL engine */
GPSACQ_API void* gpsacq_stream(gpsacq_engine* e);

/*
 * 8-bit IQ capture -> the 1-bit real-IF stream gpsacq_search() takes.  format GPSACQ_IQ_U8: rtl-sdr
 * (unsigned, offset 128, interleaved I,Q; proc_rtl_bin_for_gps.m:12-16), GPSACQ_IQ_S8: HackRF (signed;
 * proc_hackrf_bin_for_gps.m:7-11).  remove_dc: subtract the complex mean of the whole capture
 * (`y = y - mean(y)`).  mix_hz != 0: take real(y * exp(2 pi i mix_hz n / fs)) (proc_rtl...m:41),
 * else the real part.  fs <= 0: the engine's fs.  Output: ceil(n/8) bytes, sample n in bit n%8 of
 * byte n/8 (MATLAB 'ubit1'), bit = 1 where the value is <= 0.  The _device form takes device
 * pointers (IQ 16-byte aligned) and runs on the engine's stream.
 */
#define GPSACQ_IQ_U8 0
#define GPSACQ_IQ_S8 1
GPSACQ_API int gpsacq_iq8_to_bits(gpsacq_engine* e, const void* iq, size_t n_samples, int format, int remove_dc,
                       double mix_hz, double fs, uint8_t* bits_out);
GPSACQ_API int gpsacq_iq8_to_bits_device(gpsacq_engine* e, const void* d_iq, size_t n_samples, int format, int remove_dc,
                              double mix_hz, double fs, void* d_bits_out, int sync);

/*
 * Search an 8-bit IQ capture directly: README.md:83-115's rtl-sdr / HackRF flow (MATLAB conversion to a 1-bit file, then
 * gps_test on that file) as one call.  The forward transform stages each block from the IQ bytes -- mean removal, mixer,
 * sign, LO quadrature mix and bit transpose in one pass -- so the 1-bit stream is never written; results are identical to
 * gpsacq_iq8_to_bits() followed by gpsacq_search() (same per-sample arithmetic, csrc/iq_convert.hpp).  Block b starts
 * `stride` bytes into `iq` (a multiple of 16, >= 80000; 81920 = the 40960 samples one Sample() call consumes) and its
 * first 40000 samples are transformed.  With ref_quirks the stream is converted to bits first (the patch needs samples
 * 40000..40959 as bits) -- same results, one more pass.
 */
#define GPSACQ_SAMPLES_SIGN 0
#define GPSACQ_SAMPLES_REAL 1
#define GPSACQ_SAMPLES_COMPLEX 2
typedef struct gpsacq_iq8_input {
    int32_t format;          /* GPSACQ_IQ_U8 / GPSACQ_IQ_S8 */
    int32_t remove_dc;       /* subtract (mean_i, mean_q) */
    double mean_i, mean_q;   /* complex mean of the WHOLE capture in sample units (`y - mean(y)`, proc_rtl_bin_for_gps.m:17):
                                integer sums / sample count; gpsacq_iq8_accumulate_sums() for a capture streamed in pieces */
    double mix_hz;           /* 0: real part; else real(y * exp(2 pi i mix_hz n / fs))  (proc_rtl_bin_for_gps.m:41) */
    double fs;               /* sampling rate of the mixer phase; <= 0: the engine's */
    uint64_t first_sample;   /* capture sample index of the buffer's first sample: the mixer's n */
    uint64_t total_samples;  /* samples in the whole capture (bits beyond it read 0); 0: every block handed over is complete */
    int32_t multibit;        /* 0: the sign of each sample, as the scripts write it and gps_test reads it.  1: the samples keep their
                                amplitude ("direct float path"; no reference counterpart -- gps_test only takes 1-bit files): the
                                real-IF value as a float, the quadrature LO of Sample() (:143-153) applied as signs; spares the
                                1-bit quantisation loss.  2 (GPSACQ_SAMPLES_COMPLEX): the capture is at baseband already -- I + jQ
                                is what Sample() builds in fwd_buf (:149-150), e.g. the int8 +-30 file the reference's own
                                c/conv_1bit_bin_to_hackrf_bin.cpp:61-80 writes for HackRF replay, or gps_bin1bit_log2bin.m's +-100 one -- so no LO: the complex samples
                                (less the mean, turned by exp(2 pi i mix_hz n / fs) when a residual IF is named) are transformed
                                as they are; the engine's fc plays no part.  1 and 2: not with ref_quirks; on a Doppler grid finer
                                than a bin the sub-bin turn is applied to the float samples (one copy per sub-bin offset). */
    int32_t reserved;
} gpsacq_iq8_input;
GPSACQ_API int gpsacq_search_iq8(gpsacq_engine* e, const gpsacq_iq8_input* in, const void* iq, size_t n_blocks, size_t stride,
                      const gpsacq_task* tasks, size_t n_tasks, gpsacq_cell* cells, gpsacq_peak* peaks);
GPSACQ_API int gpsacq_search_iq8_device(gpsacq_engine* e, const gpsacq_iq8_input* in, const void* d_iq, size_t n_blocks, size_t stride,
                             const void* d_tasks, size_t n_tasks, void* d_cells, void* d_peaks, int sync);
/* adds the exact integer sums of I and Q (offset removed for GPSACQ_IQ_U8) over n_samples of a host buffer to sums[0..1] */
GPSACQ_API int gpsacq_iq8_accumulate_sums(gpsacq_engine* e, const void* iq, size_t n_samples, int format, int64_t sums[2]);

/*
 * Synthetic 1-bit real-IF capture generated on the device (the reference's gps_sig_gen.m writes one
 * noise-free PRN; this is the signal model of SURVEY.md section 8d): white Gaussian noise of standard
 * deviation noise_sigma plus, per satellite, amplitude * C/A chip * cos(2 pi ((fc + doppler)/fs m +
 * carrier_phase)), chips advancing at 1.023e6 (1 + doppler/L1) per second from code_phase_samples;
 * bit = (sum < 0), sample m in bit m % 8 of byte m / 8.  Uses the engine's fc and fs.  Deterministic
 * in (seed, arguments).  A search of the result reports lo_shift = round(doppler * 40000 / fs) and
 * ca_shift = (code_phase + 8 * stride * block) mod (fs / 1000) for block-sized strides.
 */
typedef struct {
    int32_t prn;                 /* 1..32 */
    float amplitude;             /* relative to noise_sigma = 1: 0.151 ~ 45 dB-Hz */
    double doppler_hz;
    double code_phase_samples;
    double carrier_phase_cycles;
} gpsacq_sat;
GPSACQ_API int gpsacq_generate(gpsacq_engine* e, uint8_t* bits_out, size_t n_bytes, const gpsacq_sat* sats, int n_sats,
                    float noise_sigma, uint64_t seed);
GPSACQ_API int gpsacq_generate_device(gpsacq_engine* e, void* d_bits_out, size_t n_bytes, const gpsacq_sat* sats, int n_sats,
                           float noise_sigma, uint64_t seed, int sync);
/* Any byte range of that stream: the n_bytes that start at sample first_sample (a multiple of 8) -- byte first_sample / 8 of what
 * gpsacq_generate() writes for the same seed and satellites, bit for bit (noise and signals are functions of the absolute sample
 * index).  One rank of a multi-GPU job generates exactly its own blocks of THE capture every world size searches. */
GPSACQ_API int gpsacq_generate_range(gpsacq_engine* e, uint8_t* bits_out, size_t n_bytes, uint64_t first_sample, const gpsacq_sat* sats,
                          int n_sats, float noise_sigma, uint64_t seed);
GPSACQ_API int gpsacq_generate_range_device(gpsacq_engine* e, void* d_bits_out, size_t n_bytes, uint64_t first_sample,
                                 const gpsacq_sat* sats, int n_sats, float noise_sigma, uint64_t seed, int sync);

/*
 * The reference's own test signal (gps_sig_gen.m:8-41, the script that wrote gps_sig_tmp.bin), generated on the device:
 * PRN `prn` (1..32), noise-free BPSK with the given navigation bits (+-1, 20 code periods each), 8 samples per chip
 * (fs = 8.184 MHz), raised-cosine shaping (MATLAB rcosine(1, 8)), carrier at fs/4 = 2.046 MHz, 1 bit per sample, LSB first.
 * Arithmetic in double in the script's own order: with the 100 bits of the bundled file (tests/golden/
 * gps_sig_tmp_databits.json; the script draws them with an unseeded rand) the output equals gps_sig_tmp.bin bit for bit.
 * n_bytes must be gpsacq_sig_bytes(n_data_bits) = ceil((n_data_bits * 20 * 1023 * 8 + 48) / 8).  Search it with
 * fc = 2.046e6, fs = 8.184e6.
 */
GPSACQ_API size_t gpsacq_sig_bytes(int n_data_bits);
GPSACQ_API int gpsacq_generate_sig(gpsacq_engine* e, int prn, const int8_t* data_bits, int n_data_bits, uint8_t* bits_out, size_t n_bytes);
/*
 * The script's OTHER output (gps_sig_gen.m:21-30, gps_sig_tmp_for_hackrf_tx.bin -- what README.md section 2.2 replays through a
 * HackRF): the same shaped baseband, the navigation-bit sequence n_repeat times over (5 in the script), times 50, as 8-bit
 * complex samples at IF 0: I = int8(round(x * 50)), Q = 0, interleaved.  The stream has gpsacq_sig_tx_samples(n_data_bits,
 * n_repeat) = n_repeat * n_data_bits * 20 * 1023 * 8 + 48 complex samples (164 MB for the script's 100 bits x 5); any range
 * [first_sample, first_sample + n_samples) of it is written to iq_out[2 * n_samples].  Search it as complex baseband:
 * gpsacq_iq8_input{format = GPSACQ_IQ_S8, remove_dc = 0, multibit = GPSACQ_SAMPLES_COMPLEX} at fs = 8.184e6 (fc plays no part).
 */
GPSACQ_API uint64_t gpsacq_sig_tx_samples(int n_data_bits, int n_repeat);
GPSACQ_API int gpsacq_generate_sig_tx(gpsacq_engine* e, int prn, const int8_t* data_bits, int n_data_bits, int n_repeat,
                           uint64_t first_sample, size_t n_samples, int8_t* iq_out);

/*
 * Acquisition hand-off record: what the tracking channel derives from a search hit
 * (c/channel.cpp:144-163).  Host arithmetic only; no device needed.
 */
typedef struct {
    double lo_dop_hz;   /* carrier Doppler estimate  = lo_shift * fs / 40000                  (:145) */
    double ca_dop_hz;   /* code-rate Doppler         = lo_dop / L1 * 1.023e6                  (:146) */
    uint32_t lo_rate;   /* carrier NCO word          = (fc  + lo_dop) / fs * 2^32             (:149) */
    uint32_t ca_rate;   /* code NCO word             = (CPS + ca_dop) / fs * 2^32             (:150) */
    int32_t ca_shift;   /* code phase after creep    = ca_shift + nearbyint(ca_dop*secs*fs/CPS) (:160) */
    uint32_t ca_pause;  /* NCO pause to align the code generator = (2*spm - ca_shift) % spm, spm = samples
                           per millisecond (the reference hard-codes 20000 / 10000 for its 10 MHz FPGA, :163) */
} gpsacq_handoff_t;
GPSACQ_API int gpsacq_handoff(const gpsacq_peak* peak, double fc, double fs, double secs_since_sample, gpsacq_handoff_t* out);
/* gpsacq_handoff() reads lo_shift in FFT bins of fs / 40000 Hz -- the reference grid.  After gpsacq_set_doppler_step(), and
 * for gpsacq_multi_search_grid() peaks, lo_shift counts grid points of gpsacq_info.doppler_step_hz: pass that step here
 * (step_hz <= 0: FFT bins), or let the engine supply its own fc, fs and current step. */
GPSACQ_API int gpsacq_handoff_step(const gpsacq_peak* peak, double fc, double fs, double step_hz, double secs_since_sample, gpsacq_handoff_t* out);
GPSACQ_API int gpsacq_handoff_engine(const gpsacq_engine* e, const gpsacq_peak* peak, double secs_since_sample, gpsacq_handoff_t* out);

/*
 * Single-process multi-GPU search of ONE capture's PRN x Doppler grid (BASELINE.json configs[4]; the reference is
 * single-threaded, c/search_offline.cpp has no counterpart).  One engine per listed device; the Doppler grid -K..+K is
 * cut into one contiguous slab per device, every device searches all tasks over its slab, packs each task's best peak
 * into a 64-bit key (snr bits << 32 | (0xFFFF - grid index) << 16 | ca_shift: integer MAX = higher SNR, ties to the
 * LOWER Doppler point like the strict '>' of :196-198) and ONE ncclAllReduce(MAX, uint64) over RCCL (xGMI) merges them.
 * devices == NULL: ordinals 0..n_devices-1.  n_devices == 1 is the degenerate case (same result as gpsacq_search's
 * peaks).  peaks[n_tasks]: snr / lo_shift / ca_shift of the merged best and its max_pwr -- the key has no room for the
 * power, so the winner's value follows in a second all-reduce(MAX, float) (the engine whose key won contributes its value,
 * the others 0).  A device may be listed more than once: engines of one GPU merge their keys on that GPU and RCCL joins the
 * distinct GPUs only.  RCCL is dlopen'ed on first use (librccl.so.1); GPSACQ_ERR_DEVICE if it is missing.  With one distinct
 * GPU no RCCL call is made -- unless GPSACQ_MULTI_FORCE_RCCL=1 is set when the handle is created: then the communicator
 * (ncclCommInitAll over one device) and both all-reduces run as a one-rank group, the same calls a multi-GPU node makes.
 * Each engine's share of a call (staging copy out of the caller's pageable buffer, upload, search, download) is enqueued by
 * its own host thread through pinned buffers, so no device waits for another one's copies.
 */
typedef struct gpsacq_multi gpsacq_multi;
GPSACQ_API int gpsacq_multi_create(const gpsacq_params* params, const int32_t* devices, int n_devices, gpsacq_multi** out);
GPSACQ_API void gpsacq_multi_destroy(gpsacq_multi* m);
GPSACQ_API int gpsacq_multi_set_doppler_step(gpsacq_multi* m, double step_hz);
GPSACQ_API int gpsacq_multi_get_info(const gpsacq_multi* m, gpsacq_info* info, int32_t* n_devices);
GPSACQ_API int gpsacq_multi_search_grid(gpsacq_multi* m, const uint8_t* bits, size_t n_blocks, size_t stride,
                             const gpsacq_task* tasks, size_t n_tasks, gpsacq_peak* peaks);
/*
 * The headline decomposition (BASELINE.json north_star) in one process: a capture of whole runs
 * (32 blocks of `stride` >= 5120 bytes each, block b against PRN b % 32 -- the SearchTask() loop, c/search_offline.cpp:237-262)
 * is cut into one contiguous range of runs per device; every device searches its runs over the full Doppler grid, reduces
 * its peaks to the best per PRN as packed keys, and ONE ncclAllReduce(MAX, uint64) of 32 keys merges them.  Outputs, either
 * may be NULL: peaks[n_runs * 32] -- every (run, PRN) peak in file order, what SearchTask() reports -- and best[32], the
 * merged per-PRN best over the whole capture (max_pwr included: the winner's value, merged next to the keys as above).
 * n_devices == 1 is the degenerate case.  The engines' Doppler windows are left as they were.
 */
GPSACQ_API int gpsacq_multi_search_blocks(gpsacq_multi* m, const uint8_t* bits, size_t n_runs, size_t stride, gpsacq_peak* peaks,
                               gpsacq_peak* best);
/* Host-side times of the last gpsacq_multi_search_* call, milliseconds: enqueue_ms = entry until every device's work and the
 * merge were enqueued (what the calling thread costs the devices; flat in the number of devices), total_ms = the whole call;
 * rccl_allreduces = ncclAllReduce groups issued by this handle so far (0 when no communicator exists).  Any pointer may be NULL. */
GPSACQ_API int gpsacq_multi_last_call_ms(const gpsacq_multi* m, double* enqueue_ms, double* total_ms, int64_t* rccl_allreduces);

/* SearchCode(): chips to clock PRN sv's generator until its G1 register reads g1 (-1 if never) */
GPSACQ_API int gpsacq_search_code(int sv, int g1);

/* Parity probes (natural bin order, interleaved re/im, 40000 complex floats each). */
GPSACQ_API int gpsacq_sample_spectrum(gpsacq_engine* e, const uint8_t* block5120, float* out);
GPSACQ_API int gpsacq_code_spectrum(gpsacq_engine* e, int sv, float* out);

#ifdef __cplusplus
}
#endif
#endif /* GPSACQ_H */
