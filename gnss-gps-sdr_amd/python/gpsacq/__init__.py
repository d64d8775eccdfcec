"""ctypes binding of libgpsacq.so (include/gpsacq.h) -- the MI355X GPS L1 C/A acquisition engine.

Thin on purpose: every number comes out of the HIP kernels behind the C ABI.  If the shared
library (built by `make lib` / __graft_entry__.build()) is missing this module raises; there
is no Python or CPU fallback for the search.

Reference interface mirrored (JiaoXianjun/GNSS-GPS-SDR, c/gps_offline.h:87-91 and the globals
FC/FS/max_fo of c/gps_offline.h:23-25): `Engine(fc, fs, max_fo)` is SearchInit(),
`Engine.search()` is the Sample()+Correlate() body of SearchTask()'s loop
(c/search_offline.cpp:239-246), `Engine.close()` is SearchFree(), `search_code()` is SearchCode().
"""
import ctypes
import os

import numpy as np

FFT_LEN = 40000
NUM_SATS = 32
BLOCK_BYTES = 5120
THRESHOLD = 25.0  # c/search_offline.cpp:248
STAMP_SLOTS = 512  # GPSACQ_STAMP_SLOTS: one cycle-counter slot per (XCD, shader engine, compute unit)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GPSACQ_LIB") or os.path.normpath(os.path.join(_HERE, "..", "..", "lib", "libgpsacq.so"))

CELL_DTYPE = np.dtype([("max_pwr", "<f4"), ("max_i", "<i4"), ("tot_pwr", "<f4"), ("snr", "<f4")])
PEAK_DTYPE = np.dtype([("snr", "<f4"), ("lo_shift", "<i4"), ("ca_shift", "<i4"), ("max_pwr", "<f4")])
TASK_DTYPE = np.dtype([("block", "<i4"), ("prn", "<i4")])


class Params(ctypes.Structure):
    _fields_ = [("fc", ctypes.c_double), ("fs", ctypes.c_double), ("max_fo", ctypes.c_double),
                ("device", ctypes.c_int32), ("ref_quirks", ctypes.c_int32)]


class Info(ctypes.Structure):
    _fields_ = [("fft_len", ctypes.c_int32), ("dmax", ctypes.c_int32), ("num_doppler", ctypes.c_int32),
                ("first_doppler", ctypes.c_int32), ("num_lags", ctypes.c_int32), ("acc_columns", ctypes.c_int32), ("device", ctypes.c_int32),
                ("compute_units", ctypes.c_int32), ("device_name", ctypes.c_char * 64),
                ("doppler_sub", ctypes.c_int32), ("doppler_stride", ctypes.c_int32), ("num_doppler_total", ctypes.c_int32),
                ("first_doppler_total", ctypes.c_int32), ("doppler_step_hz", ctypes.c_double)]


class Timing(ctypes.Structure):
    _fields_ = [("ms_total", ctypes.c_float), ("ms_sample", ctypes.c_float), ("ms_correlate", ctypes.c_float),
                ("ms_peaks", ctypes.c_float), ("correlate_launches", ctypes.c_int32), ("cells", ctypes.c_int64)]


class Handoff(ctypes.Structure):
    _fields_ = [("lo_dop_hz", ctypes.c_double), ("ca_dop_hz", ctypes.c_double), ("lo_rate", ctypes.c_uint32),
                ("ca_rate", ctypes.c_uint32), ("ca_shift", ctypes.c_int32), ("ca_pause", ctypes.c_uint32)]


class Iq8Input(ctypes.Structure):
    _fields_ = [("format", ctypes.c_int32), ("remove_dc", ctypes.c_int32), ("mean_i", ctypes.c_double), ("mean_q", ctypes.c_double),
                ("mix_hz", ctypes.c_double), ("fs", ctypes.c_double), ("first_sample", ctypes.c_uint64), ("total_samples", ctypes.c_uint64),
                ("multibit", ctypes.c_int32), ("reserved", ctypes.c_int32)]


class Sat(ctypes.Structure):
    _fields_ = [("prn", ctypes.c_int32), ("amplitude", ctypes.c_float), ("doppler_hz", ctypes.c_double),
                ("code_phase_samples", ctypes.c_double), ("carrier_phase_cycles", ctypes.c_double)]


EXPORTS = ["gpsacq_generate", "gpsacq_generate_device", "gpsacq_generate_range", "gpsacq_generate_range_device", "gpsacq_generate_sig", "gpsacq_sig_bytes", "gpsacq_handoff", "gpsacq_iq8_to_bits", "gpsacq_iq8_to_bits_device", "gpsacq_create", "gpsacq_destroy", "gpsacq_last_error", "gpsacq_get_info", "gpsacq_search",
           "gpsacq_search_device", "gpsacq_set_doppler_window", "gpsacq_set_cell_handout", "gpsacq_set_doppler_step", "gpsacq_set_noncoherent", "gpsacq_set_creep_compensation", "gpsacq_set_block_alignment", "gpsacq_aligned_stride", "gpsacq_synchronize", "gpsacq_last_timing", "gpsacq_timing_ago", "gpsacq_stream", "gpsacq_search_code",
           "gpsacq_sample_spectrum", "gpsacq_code_spectrum", "gpsacq_multi_create", "gpsacq_multi_destroy",
           "gpsacq_multi_set_doppler_step", "gpsacq_multi_get_info", "gpsacq_multi_search_grid", "gpsacq_multi_search_blocks",
           "gpsacq_pipe_buffer", "gpsacq_pipe_submit", "gpsacq_pipe_collect", "gpsacq_search_iq8", "gpsacq_search_iq8_device",
           "gpsacq_iq8_accumulate_sums", "gpsacq_handoff_step", "gpsacq_handoff_engine", "gpsacq_reserve", "gpsacq_multi_last_call_ms",
           "gpsacq_sig_tx_samples", "gpsacq_generate_sig_tx", "gpsacq_peak_keys_device", "gpsacq_cycle_stamp_device"]

_lib = None


def _preload_torch():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so (SONAME libamdhip64.so.7, the same as
    /opt/rocm's).  Loaded first, torch's copy satisfies libgpsacq's NEEDED libamdhip64.so.7 by SONAME
    and the process has ONE HIP runtime (streams and events are then interchangeable, which
    bench.py relies on).  Loaded second, torch asks for the file name `libamdhip64.so`, gets its own
    second copy next to /opt/rocm's, and its CUDA initialisation finds no GPU (measured on the
    MI355X box, both orders).  So: if torch is installed and not yet imported, import it before the
    dlopen.  GPSACQ_NO_TORCH_PRELOAD=1 skips this."""
    import importlib.util
    import sys
    if "torch" in sys.modules or os.environ.get("GPSACQ_NO_TORCH_PRELOAD"):
        return
    try:
        if importlib.util.find_spec("torch") is not None:
            import torch  # noqa: F401
    except Exception:
        pass


def load_library(path=None):
    """dlopen libgpsacq.so and declare the prototypes.  Raises OSError if it is not built."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise OSError(f"{p} not found: build it with `make lib` (hipcc --offload-arch=gfx950); "
                      "gpsacq has no CPU fallback")
    _preload_torch()
    lib = ctypes.CDLL(p)
    vp, sz = ctypes.c_void_p, ctypes.c_size_t
    lib.gpsacq_create.argtypes = [ctypes.POINTER(Params), ctypes.POINTER(vp)]
    lib.gpsacq_create.restype = ctypes.c_int
    lib.gpsacq_destroy.argtypes = [vp]
    lib.gpsacq_destroy.restype = None
    lib.gpsacq_last_error.argtypes = []
    lib.gpsacq_last_error.restype = ctypes.c_char_p
    lib.gpsacq_get_info.argtypes = [vp, ctypes.POINTER(Info)]
    lib.gpsacq_get_info.restype = ctypes.c_int
    lib.gpsacq_search.argtypes = [vp, vp, sz, sz, vp, sz, vp, vp]
    lib.gpsacq_search.restype = ctypes.c_int
    lib.gpsacq_search_device.argtypes = [vp, vp, sz, sz, vp, sz, vp, vp, ctypes.c_int]
    lib.gpsacq_search_device.restype = ctypes.c_int
    lib.gpsacq_set_doppler_window.argtypes = [vp, ctypes.c_int, ctypes.c_int]
    lib.gpsacq_set_doppler_window.restype = ctypes.c_int
    lib.gpsacq_set_cell_handout.argtypes = [vp, ctypes.c_int]
    lib.gpsacq_set_cell_handout.restype = ctypes.c_int
    lib.gpsacq_set_doppler_step.argtypes = [vp, ctypes.c_double]
    lib.gpsacq_set_doppler_step.restype = ctypes.c_int
    lib.gpsacq_set_noncoherent.argtypes = [vp, ctypes.c_int, ctypes.c_int]
    lib.gpsacq_set_noncoherent.restype = ctypes.c_int
    lib.gpsacq_aligned_stride.argtypes = [vp]
    lib.gpsacq_aligned_stride.restype = ctypes.c_int
    lib.gpsacq_synchronize.argtypes = [vp]
    lib.gpsacq_synchronize.restype = ctypes.c_int
    lib.gpsacq_last_timing.argtypes = [vp, ctypes.POINTER(Timing)]
    lib.gpsacq_last_timing.restype = ctypes.c_int
    lib.gpsacq_set_creep_compensation.argtypes = [vp, ctypes.c_int]
    lib.gpsacq_set_creep_compensation.restype = ctypes.c_int
    lib.gpsacq_set_block_alignment.argtypes = [vp, ctypes.c_int]
    lib.gpsacq_set_block_alignment.restype = ctypes.c_int
    lib.gpsacq_timing_ago.argtypes = [vp, ctypes.c_int, ctypes.POINTER(Timing)]
    lib.gpsacq_timing_ago.restype = ctypes.c_int
    lib.gpsacq_stream.argtypes = [vp]
    lib.gpsacq_stream.restype = ctypes.c_void_p
    lib.gpsacq_iq8_to_bits.argtypes = [vp, vp, sz, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_double, vp]
    lib.gpsacq_iq8_to_bits.restype = ctypes.c_int
    lib.gpsacq_iq8_to_bits_device.argtypes = [vp, vp, sz, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_double, vp, ctypes.c_int]
    lib.gpsacq_iq8_to_bits_device.restype = ctypes.c_int
    lib.gpsacq_generate.argtypes = [vp, vp, sz, ctypes.POINTER(Sat), ctypes.c_int, ctypes.c_float, ctypes.c_uint64]
    lib.gpsacq_generate.restype = ctypes.c_int
    lib.gpsacq_generate_device.argtypes = [vp, vp, sz, ctypes.POINTER(Sat), ctypes.c_int, ctypes.c_float, ctypes.c_uint64, ctypes.c_int]
    lib.gpsacq_generate_device.restype = ctypes.c_int
    lib.gpsacq_generate_range.argtypes = [vp, vp, sz, ctypes.c_uint64, ctypes.POINTER(Sat), ctypes.c_int, ctypes.c_float, ctypes.c_uint64]
    lib.gpsacq_generate_range.restype = ctypes.c_int
    lib.gpsacq_generate_range_device.argtypes = [vp, vp, sz, ctypes.c_uint64, ctypes.POINTER(Sat), ctypes.c_int, ctypes.c_float, ctypes.c_uint64, ctypes.c_int]
    lib.gpsacq_generate_range_device.restype = ctypes.c_int
    lib.gpsacq_sig_bytes.argtypes = [ctypes.c_int]
    lib.gpsacq_sig_bytes.restype = ctypes.c_size_t
    lib.gpsacq_generate_sig.argtypes = [vp, ctypes.c_int, vp, ctypes.c_int, vp, sz]
    lib.gpsacq_generate_sig.restype = ctypes.c_int
    lib.gpsacq_handoff.argtypes = [vp, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.POINTER(Handoff)]
    lib.gpsacq_handoff.restype = ctypes.c_int
    lib.gpsacq_search_code.argtypes = [ctypes.c_int, ctypes.c_int]
    lib.gpsacq_search_code.restype = ctypes.c_int
    lib.gpsacq_sample_spectrum.argtypes = [vp, vp, vp]
    lib.gpsacq_sample_spectrum.restype = ctypes.c_int
    lib.gpsacq_code_spectrum.argtypes = [vp, ctypes.c_int, vp]
    lib.gpsacq_code_spectrum.restype = ctypes.c_int
    lib.gpsacq_multi_create.argtypes = [ctypes.POINTER(Params), vp, ctypes.c_int, ctypes.POINTER(vp)]
    lib.gpsacq_multi_create.restype = ctypes.c_int
    lib.gpsacq_multi_destroy.argtypes = [vp]
    lib.gpsacq_multi_destroy.restype = None
    lib.gpsacq_multi_set_doppler_step.argtypes = [vp, ctypes.c_double]
    lib.gpsacq_multi_set_doppler_step.restype = ctypes.c_int
    lib.gpsacq_multi_get_info.argtypes = [vp, ctypes.POINTER(Info), ctypes.POINTER(ctypes.c_int32)]
    lib.gpsacq_multi_get_info.restype = ctypes.c_int
    lib.gpsacq_multi_search_grid.argtypes = [vp, vp, sz, sz, vp, sz, vp]
    lib.gpsacq_multi_search_grid.restype = ctypes.c_int
    lib.gpsacq_multi_search_blocks.argtypes = [vp, vp, sz, sz, vp, vp]
    lib.gpsacq_multi_search_blocks.restype = ctypes.c_int
    lib.gpsacq_pipe_buffer.argtypes = [vp, ctypes.c_int, sz]
    lib.gpsacq_pipe_buffer.restype = ctypes.c_void_p
    lib.gpsacq_pipe_submit.argtypes = [vp, ctypes.c_int, sz, sz, ctypes.POINTER(Iq8Input)]
    lib.gpsacq_pipe_submit.restype = ctypes.c_int
    lib.gpsacq_pipe_collect.argtypes = [vp, ctypes.c_int, ctypes.POINTER(vp), ctypes.POINTER(sz)]
    lib.gpsacq_pipe_collect.restype = ctypes.c_int
    lib.gpsacq_search_iq8.argtypes = [vp, ctypes.POINTER(Iq8Input), vp, sz, sz, vp, sz, vp, vp]
    lib.gpsacq_search_iq8.restype = ctypes.c_int
    lib.gpsacq_search_iq8_device.argtypes = [vp, ctypes.POINTER(Iq8Input), vp, sz, sz, vp, sz, vp, vp, ctypes.c_int]
    lib.gpsacq_search_iq8_device.restype = ctypes.c_int
    lib.gpsacq_iq8_accumulate_sums.argtypes = [vp, vp, sz, ctypes.c_int, ctypes.POINTER(ctypes.c_int64)]
    lib.gpsacq_iq8_accumulate_sums.restype = ctypes.c_int
    lib.gpsacq_handoff_step.argtypes = [vp, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.POINTER(Handoff)]
    lib.gpsacq_handoff_step.restype = ctypes.c_int
    lib.gpsacq_handoff_engine.argtypes = [vp, vp, ctypes.c_double, ctypes.POINTER(Handoff)]
    lib.gpsacq_handoff_engine.restype = ctypes.c_int
    lib.gpsacq_reserve.argtypes = [vp, sz]
    lib.gpsacq_reserve.restype = ctypes.c_int
    lib.gpsacq_sig_tx_samples.argtypes = [ctypes.c_int, ctypes.c_int]
    lib.gpsacq_sig_tx_samples.restype = ctypes.c_uint64
    lib.gpsacq_generate_sig_tx.argtypes = [vp, ctypes.c_int, vp, ctypes.c_int, ctypes.c_int, ctypes.c_uint64, sz, vp]
    lib.gpsacq_generate_sig_tx.restype = ctypes.c_int
    lib.gpsacq_peak_keys_device.argtypes = [vp, vp, sz, ctypes.c_int, vp, ctypes.c_int]
    lib.gpsacq_peak_keys_device.restype = ctypes.c_int
    lib.gpsacq_cycle_stamp_device.argtypes = [vp, vp, ctypes.c_int]
    lib.gpsacq_cycle_stamp_device.restype = ctypes.c_int
    lib.gpsacq_multi_last_call_ms.argtypes = [vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int64)]
    lib.gpsacq_multi_last_call_ms.restype = ctypes.c_int
    if path is None:
        _lib = lib
    return lib


class GpsAcqError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"gpsacq error {code}: {msg}")
        self.code = code


def _check(lib, rc):
    if rc != 0:
        raise GpsAcqError(rc, lib.gpsacq_last_error().decode(errors="replace"))


def handoff(peak, fc, fs, secs_since_sample=0.0, step_hz=0.0):
    """CHANNEL::Start()'s NCO set-up from a search hit (c/channel.cpp:134-163).  peak: a PEAK_DTYPE record.
    step_hz: what one unit of lo_shift is worth -- 0 for the reference grid (FFT bins of fs/40000), else the engine's
    doppler_step_hz after set_doppler_step() (Engine.handoff() passes it by itself)."""
    lib = load_library()
    pk = np.zeros(1, dtype=PEAK_DTYPE)
    pk[0] = peak
    h = Handoff()
    _check(lib, lib.gpsacq_handoff_step(pk.ctypes.data_as(ctypes.c_void_p), float(fc), float(fs), float(step_hz), float(secs_since_sample), ctypes.byref(h)))
    return {k: getattr(h, k) for k, _ in Handoff._fields_}


def search_code(sv, g1):
    """SearchCode(), c/search_offline.cpp:205-209."""
    return load_library().gpsacq_search_code(int(sv), int(g1))


class Engine:
    """SearchInit() for one (FC, FS, max_fo); owns the device state."""

    def __init__(self, fc, fs, max_fo=5000.0, device=0, ref_quirks=False):
        self._lib = load_library()
        self._h = ctypes.c_void_p()
        prm = Params(float(fc), float(fs), float(max_fo), int(device), 1 if ref_quirks else 0)
        _check(self._lib, self._lib.gpsacq_create(ctypes.byref(prm), ctypes.byref(self._h)))
        self._refresh_info()
        self.fc, self.fs, self.max_fo = float(fc), float(fs), float(max_fo)
        self._quirks = bool(ref_quirks)
        # run-time hand-out of the correlate kernel's cells (gpsacq_set_cell_handout): on unless the environment said 0 at gpsacq_create
        self.cell_handout = os.environ.get("GPSACQ_CORR_PERSIST") != "0"

    def _refresh_info(self):
        info = Info()
        _check(self._lib, self._lib.gpsacq_get_info(self._h, ctypes.byref(info)))
        self.dmax = info.dmax
        self.num_doppler = info.num_doppler
        self.first_doppler = info.first_doppler
        self.num_lags = info.num_lags
        self.acc_columns = info.acc_columns
        self.device = info.device
        self.compute_units = info.compute_units
        self.device_name = info.device_name.decode(errors="replace")
        self.doppler_sub = info.doppler_sub
        self.doppler_stride = info.doppler_stride
        self.num_doppler_total = info.num_doppler_total
        self.first_doppler_total = info.first_doppler_total
        self.kmax = -info.first_doppler_total  # grid points run -kmax..+kmax (= dmax on the reference grid)
        self.doppler_step_hz = info.doppler_step_hz

    def set_doppler_window(self, first_bin, n_bins):
        """Search only bins first_bin .. first_bin+n_bins-1 (multi-GPU Doppler-slab sharding)."""
        _check(self._lib, self._lib.gpsacq_set_doppler_window(self._h, int(first_bin), int(n_bins)))
        self._refresh_info()

    def set_cell_handout(self, on):
        """Run-time hand-out of the correlate kernel's cells to persistent workgroups (default on) or one workgroup per cell
        (gpsacq_set_cell_handout): the same cells bit for bit, a different time."""
        _check(self._lib, self._lib.gpsacq_set_cell_handout(self._h, 1 if on else 0))
        self.cell_handout = bool(on)

    def set_doppler_step(self, step_hz):
        """Doppler grid step in Hz: finer than fs/40000 through sub-bin spectra, coarser through a bin stride
        (include/gpsacq.h).  lo_shift / windows / cells columns then count grid points of doppler_step_hz."""
        _check(self._lib, self._lib.gpsacq_set_doppler_step(self._h, float(step_hz)))
        self._refresh_info()

    def set_noncoherent(self, n_acc, block_step=1):
        """Sum |IFFT|^2 over n_acc blocks (block_step apart) per cell before the peak scan; 1 = reference."""
        _check(self._lib, self._lib.gpsacq_set_noncoherent(self._h, int(n_acc), int(block_step)))

    def set_creep_compensation(self, on=True):
        """Non-coherent mode: re-align each accumulated block by the code creep of the cell's Doppler bin."""
        _check(self._lib, self._lib.gpsacq_set_creep_compensation(self._h, 1 if on else 0))

    def set_block_alignment(self, on=True):
        """Non-coherent mode: re-align each accumulated block by the code phase between block starts (any stride)."""
        _check(self._lib, self._lib.gpsacq_set_block_alignment(self._h, 1 if on else 0))

    def aligned_stride(self):
        """Bytes between block starts that keep lags aligned for non-coherent sums (whole C/A periods)."""
        return self._lib.gpsacq_aligned_stride(self._h)

    def close(self):
        if self._h:
            self._lib.gpsacq_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def handoff(self, peak, secs_since_sample=0.0):
        """Hand-off record of a hit of THIS engine: its fc, fs and current Doppler grid step."""
        pk = np.zeros(1, dtype=PEAK_DTYPE)
        pk[0] = peak
        h = Handoff()
        _check(self._lib, self._lib.gpsacq_handoff_engine(self._h, pk.ctypes.data_as(ctypes.c_void_p), float(secs_since_sample), ctypes.byref(h)))
        return {k: getattr(h, k) for k, _ in Handoff._fields_}

    # ---- 8-bit IQ capture searched directly (no 1-bit intermediate) -------------------------
    @staticmethod
    def iq8_input(signed=False, remove_dc=True, mean=(0.0, 0.0), mix_hz=0.0, fs=0.0, first_sample=0, total_samples=0, multibit=False):
        return Iq8Input(1 if signed else 0, 1 if remove_dc else 0, float(mean[0]), float(mean[1]), float(mix_hz), float(fs),
                        int(first_sample), int(total_samples), int(multibit), 0)  # multibit: False/0 sign, True/1 real IF, 2 complex baseband

    def iq8_mean(self, iq, signed=False, chunk_samples=1 << 22):
        """Complex mean of a whole 8-bit IQ capture as (mean_i, mean_q): exact integer sums on the device, in pieces."""
        buf = np.ascontiguousarray(np.asarray(iq).view(np.uint8).ravel())
        n = buf.size // 2
        sums = (ctypes.c_int64 * 2)(0, 0)
        for s0 in range(0, n, chunk_samples):
            m = min(chunk_samples, n - s0)
            part = buf[2 * s0:2 * (s0 + m)]
            _check(self._lib, self._lib.gpsacq_iq8_accumulate_sums(self._h, part.ctypes.data_as(ctypes.c_void_p), m, 1 if signed else 0, sums))
        return sums[0] / n, sums[1] / n

    def search_iq8(self, iq, inp, tasks=None, stride=16 * BLOCK_BYTES, want_cells=True):
        """Search interleaved 8-bit I,Q bytes directly (gpsacq_search_iq8): blocks `stride` bytes apart (81920 = the 40960
        samples of one Sample() call).  inp: Engine.iq8_input(...)."""
        buf = np.ascontiguousarray(np.asarray(iq).view(np.uint8).ravel())
        need = 16 * (BLOCK_BYTES if self._quirks else 5000)
        n_blocks = (buf.size - min(stride, need)) // stride + 1 if buf.size >= min(stride, need) else 0
        if n_blocks <= 0:
            raise ValueError("capture shorter than one block")
        if tasks is None:
            n_tasks, tptr = n_blocks, None
        else:
            t = np.ascontiguousarray(np.asarray(tasks, dtype=np.int32).reshape(-1, 2))
            n_tasks, tptr = t.shape[0], t.ctypes.data_as(ctypes.c_void_p)
        cells = np.zeros((n_tasks, self.num_doppler), dtype=CELL_DTYPE) if want_cells else None
        peaks = np.zeros(n_tasks, dtype=PEAK_DTYPE)
        _check(self._lib, self._lib.gpsacq_search_iq8(
            self._h, ctypes.byref(inp), buf.ctypes.data_as(ctypes.c_void_p), n_blocks, stride, tptr, n_tasks,
            cells.ctypes.data_as(ctypes.c_void_p) if want_cells else None, peaks.ctypes.data_as(ctypes.c_void_p)))
        return cells, peaks

    def search_iq8_device(self, d_iq_ptr, inp, n_blocks, d_peaks_ptr, stride=16 * BLOCK_BYTES, sync=True):
        _check(self._lib, self._lib.gpsacq_search_iq8_device(self._h, ctypes.byref(inp), d_iq_ptr, n_blocks, stride, None, n_blocks,
                                                             None, d_peaks_ptr, 1 if sync else 0))

    # ---- pipelined host-buffer searches (gpsacq_pipe_*) --------------------------------------
    def pipe_buffer(self, slot, nbytes):
        """The slot's pinned staging buffer as a writable uint8 array of nbytes."""
        ptr = self._lib.gpsacq_pipe_buffer(self._h, int(slot), int(nbytes))
        if not ptr:
            raise GpsAcqError(-1, self._lib.gpsacq_last_error().decode(errors="replace"))
        return np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ctypes.c_uint8)), shape=(int(nbytes),))

    def pipe_submit(self, slot, n_blocks, stride=BLOCK_BYTES, iq=None):
        _check(self._lib, self._lib.gpsacq_pipe_submit(self._h, int(slot), int(n_blocks), int(stride), ctypes.byref(iq) if iq is not None else None))

    def pipe_collect(self, slot):
        p, n = ctypes.c_void_p(), ctypes.c_size_t()
        _check(self._lib, self._lib.gpsacq_pipe_collect(self._h, int(slot), ctypes.byref(p), ctypes.byref(n)))
        arr = np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_uint8)), shape=(n.value * PEAK_DTYPE.itemsize,))
        return arr.view(PEAK_DTYPE).copy()

    # ---- host-buffer path ----------------------------------------------------------------
    def search(self, bits, tasks=None, stride=BLOCK_BYTES, want_cells=True, n_tasks=None):
        """bits: bytes-like / uint8 array holding whole 5120-byte blocks.  tasks: None for the
        reference schedule (block t against PRN t % 32) or an array of (block, prn) pairs.
        Returns (cells[n_tasks, num_doppler] or None, peaks[n_tasks])."""
        buf = np.ascontiguousarray(np.frombuffer(bits, dtype=np.uint8) if not isinstance(bits, np.ndarray) else bits.view(np.uint8))
        n_blocks = (buf.size - min(stride, BLOCK_BYTES)) // stride + 1 if buf.size >= min(stride, BLOCK_BYTES) else 0
        if n_blocks <= 0:
            raise ValueError("capture shorter than one block")
        if tasks is None:
            n_tasks, tptr = (n_blocks if n_tasks is None else int(n_tasks)), None
        else:
            t = np.ascontiguousarray(np.asarray(tasks, dtype=np.int32).reshape(-1, 2))
            n_tasks, tptr = t.shape[0], t.ctypes.data_as(ctypes.c_void_p)
        cells = np.zeros((n_tasks, self.num_doppler), dtype=CELL_DTYPE) if want_cells else None
        peaks = np.zeros(n_tasks, dtype=PEAK_DTYPE)
        _check(self._lib, self._lib.gpsacq_search(
            self._h, buf.ctypes.data_as(ctypes.c_void_p), n_blocks, stride, tptr, n_tasks,
            cells.ctypes.data_as(ctypes.c_void_p) if want_cells else None, peaks.ctypes.data_as(ctypes.c_void_p)))
        return cells, peaks

    # ---- device-buffer path (torch tensors on this engine's device) ----------------------
    def search_device(self, d_bits_ptr, n_blocks, d_peaks_ptr, stride=BLOCK_BYTES, d_tasks_ptr=None, n_tasks=None,
                      d_cells_ptr=None, sync=True):
        n_tasks = n_blocks if n_tasks is None else n_tasks
        _check(self._lib, self._lib.gpsacq_search_device(self._h, d_bits_ptr, n_blocks, stride, d_tasks_ptr, n_tasks,
                                                         d_cells_ptr, d_peaks_ptr, 1 if sync else 0))

    def peak_keys_device(self, d_peaks_ptr, n_peaks, d_keys_ptr, per_prn=True, sync=False):
        """gpsacq_peak_keys_device: the multi-GPU merge keys of a device search's peaks, on the engine's stream.  per_prn: 32 keys
        (best per PRN, reference schedule); else one per peak.  d_keys: int64 / uint64 device memory."""
        _check(self._lib, self._lib.gpsacq_peak_keys_device(self._h, d_peaks_ptr if n_peaks else None, int(n_peaks), 1 if per_prn else 0,
                                                            d_keys_ptr, 1 if sync else 0))

    def cycle_stamp_device(self, d_stamp_ptr, sync=False):
        """gpsacq_cycle_stamp_device: every compute unit's shader-cycle counter written to d_stamp[xcc << 6 | se << 4 | cu]
        (STAMP_SLOTS x 8 bytes of zeroed device memory), on the engine's stream."""
        _check(self._lib, self._lib.gpsacq_cycle_stamp_device(self._h, d_stamp_ptr, 1 if sync else 0))

    def synchronize(self):
        _check(self._lib, self._lib.gpsacq_synchronize(self._h))

    def reserve(self, n_blocks):
        """Scratch (and the cached reference schedule) for batches of up to n_blocks blocks, once (gpsacq_reserve)."""
        _check(self._lib, self._lib.gpsacq_reserve(self._h, int(n_blocks)))

    def last_timing(self, n_back=0):
        """Stage times (ms) of the search n_back calls ago (0 = the last one); waits for that search only."""
        t = Timing()
        _check(self._lib, self._lib.gpsacq_timing_ago(self._h, int(n_back), ctypes.byref(t)))
        return {k: getattr(t, k) for k, _ in Timing._fields_}

    @property
    def stream_ptr(self):
        """The engine's hipStream_t as an integer (e.g. for torch.cuda.ExternalStream)."""
        return int(self._lib.gpsacq_stream(self._h) or 0)

    # ---- synthetic captures ---------------------------------------------------------------
    @staticmethod
    def _sats(sats):
        arr = (Sat * max(1, len(sats)))()
        for i, (prn, amp, dop, ca, ph) in enumerate(sats):
            arr[i] = Sat(int(prn), float(amp), float(dop), float(ca), float(ph))
        return arr

    def generate(self, n_bytes, sats=(), noise_sigma=1.0, seed=1, first_sample=0):
        """Synthetic 1-bit real-IF capture made on the device: sats = [(prn, amplitude, doppler_hz,
        code_phase_samples, carrier_phase_cycles), ...] on top of white noise (gps_sig_gen.m's role).  first_sample (a multiple
        of 8): the n_bytes that start there in the stream -- any range of one capture, bit for bit."""
        out = np.zeros(int(n_bytes), dtype=np.uint8)
        _check(self._lib, self._lib.gpsacq_generate_range(self._h, out.ctypes.data_as(ctypes.c_void_p), int(n_bytes), int(first_sample),
                                                          self._sats(sats), len(sats), float(noise_sigma), int(seed)))
        return out

    def generate_sig(self, prn, data_bits):
        """gps_sig_gen.m's signal on the device: PRN `prn`, navigation bits +-1 (20 code periods each), 8.184 Msps,
        IF 2.046 MHz, raised-cosine BPSK, 1 bit per sample.  Returns the packed bytes."""
        d = np.ascontiguousarray(np.asarray(data_bits, dtype=np.int8))
        n = self._lib.gpsacq_sig_bytes(int(d.size))
        out = np.zeros(n, dtype=np.uint8)
        _check(self._lib, self._lib.gpsacq_generate_sig(self._h, int(prn), d.ctypes.data_as(ctypes.c_void_p), int(d.size),
                                                        out.ctypes.data_as(ctypes.c_void_p), n))
        return out

    def generate_sig_tx(self, prn, data_bits, n_repeat=5, first_sample=0, n_samples=None):
        """gps_sig_gen.m:21-30 on the device: complex samples [first_sample, first_sample + n_samples) of the script's HackRF
        transmit file (int8 I = round(50 x), Q = 0, interleaved).  Returns an int8 array of 2 * n_samples."""
        d = np.ascontiguousarray(np.asarray(data_bits, dtype=np.int8))
        total = int(self._lib.gpsacq_sig_tx_samples(int(d.size), int(n_repeat)))
        n = total - int(first_sample) if n_samples is None else int(n_samples)
        out = np.zeros(2 * n, dtype=np.int8)
        _check(self._lib, self._lib.gpsacq_generate_sig_tx(self._h, int(prn), d.ctypes.data_as(ctypes.c_void_p), int(d.size), int(n_repeat),
                                                           int(first_sample), n, out.ctypes.data_as(ctypes.c_void_p)))
        return out

    def generate_device(self, d_bits_ptr, n_bytes, sats=(), noise_sigma=1.0, seed=1, sync=True, first_sample=0):
        _check(self._lib, self._lib.gpsacq_generate_range_device(self._h, d_bits_ptr, int(n_bytes), int(first_sample), self._sats(sats),
                                                                 len(sats), float(noise_sigma), int(seed), 1 if sync else 0))

    # ---- 8-bit IQ ingestion --------------------------------------------------------------
    def iq8_to_bits(self, iq, signed=False, remove_dc=True, mix_hz=0.0, fs=0.0):
        """proc_rtl_bin_for_gps.m / proc_hackrf_bin_for_gps.m on the device: interleaved 8-bit I,Q
        (uint8 offset-128 rtl-sdr, or int8 HackRF with signed=True) -> packed 1-bit real-IF bytes."""
        buf = np.ascontiguousarray(np.asarray(iq).view(np.uint8).ravel())
        n = buf.size // 2
        out = np.zeros((n + 7) // 8, dtype=np.uint8)
        _check(self._lib, self._lib.gpsacq_iq8_to_bits(self._h, buf.ctypes.data_as(ctypes.c_void_p), n, 1 if signed else 0,
                                                       1 if remove_dc else 0, float(mix_hz), float(fs),
                                                       out.ctypes.data_as(ctypes.c_void_p)))
        return out

    # ---- parity probes -------------------------------------------------------------------
    def sample_spectrum(self, block):
        b = np.ascontiguousarray(np.frombuffer(block, dtype=np.uint8)[:BLOCK_BYTES])
        if b.size < BLOCK_BYTES:
            raise ValueError("need 5120 bytes")
        out = np.zeros(2 * FFT_LEN, dtype=np.float32)
        _check(self._lib, self._lib.gpsacq_sample_spectrum(self._h, b.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p)))
        return out.view(np.complex64)

    def code_spectrum(self, sv):
        out = np.zeros(2 * FFT_LEN, dtype=np.float32)
        _check(self._lib, self._lib.gpsacq_code_spectrum(self._h, int(sv), out.ctypes.data_as(ctypes.c_void_p)))
        return out.view(np.complex64)


class MultiEngine:
    """gpsacq_multi_*: one capture's PRN x Doppler grid cut into Doppler slabs over several GPUs of this process, the
    per-task peaks merged by one RCCL all-reduce(MAX) of packed keys (include/gpsacq.h)."""

    def __init__(self, fc, fs, max_fo, devices=(0,)):
        self._lib = load_library()
        self._h = ctypes.c_void_p()
        prm = Params(float(fc), float(fs), float(max_fo), 0, 0)
        dev = (ctypes.c_int32 * len(devices))(*[int(d) for d in devices])
        _check(self._lib, self._lib.gpsacq_multi_create(ctypes.byref(prm), dev, len(devices), ctypes.byref(self._h)))
        self._refresh()

    def _refresh(self):
        info, n = Info(), ctypes.c_int32()
        _check(self._lib, self._lib.gpsacq_multi_get_info(self._h, ctypes.byref(info), ctypes.byref(n)))
        self.n_devices = n.value
        self.num_doppler_total = info.num_doppler_total
        self.kmax = -info.first_doppler_total
        self.doppler_step_hz = info.doppler_step_hz

    def set_doppler_step(self, step_hz):
        _check(self._lib, self._lib.gpsacq_multi_set_doppler_step(self._h, float(step_hz)))
        self._refresh()

    def search_grid(self, bits, tasks, stride=BLOCK_BYTES):
        buf = np.ascontiguousarray(np.frombuffer(bits, dtype=np.uint8) if not isinstance(bits, np.ndarray) else bits.view(np.uint8))
        n_blocks = (buf.size - min(stride, BLOCK_BYTES)) // stride + 1
        t = np.ascontiguousarray(np.asarray(tasks, dtype=np.int32).reshape(-1, 2))
        peaks = np.zeros(t.shape[0], dtype=PEAK_DTYPE)
        _check(self._lib, self._lib.gpsacq_multi_search_grid(self._h, buf.ctypes.data_as(ctypes.c_void_p), n_blocks, stride,
                                                             t.ctypes.data_as(ctypes.c_void_p), t.shape[0], peaks.ctypes.data_as(ctypes.c_void_p)))
        return peaks

    def search_blocks(self, bits, stride=BLOCK_BYTES):
        """gpsacq_multi_search_blocks: whole runs of the reference schedule split over the devices.  Returns
        (peaks[n_runs * 32] in file order, best[32] = per-PRN best after the all-reduce)."""
        buf = np.ascontiguousarray(np.frombuffer(bits, dtype=np.uint8) if not isinstance(bits, np.ndarray) else bits.view(np.uint8))
        n_blocks = (buf.size - BLOCK_BYTES) // stride + 1 if buf.size >= BLOCK_BYTES else 0
        n_runs = n_blocks // NUM_SATS
        if n_runs <= 0:
            raise ValueError("capture shorter than one run of 32 blocks")
        peaks = np.zeros(n_runs * NUM_SATS, dtype=PEAK_DTYPE)
        best = np.zeros(NUM_SATS, dtype=PEAK_DTYPE)
        _check(self._lib, self._lib.gpsacq_multi_search_blocks(self._h, buf.ctypes.data_as(ctypes.c_void_p), n_runs, stride,
                                                               peaks.ctypes.data_as(ctypes.c_void_p), best.ctypes.data_as(ctypes.c_void_p)))
        return peaks, best

    def last_call_ms(self):
        """Host-side times of the last search_* call: {"enqueue_ms", "total_ms", "rccl_allreduces"} (gpsacq_multi_last_call_ms)."""
        a, b, c = ctypes.c_double(), ctypes.c_double(), ctypes.c_int64()
        _check(self._lib, self._lib.gpsacq_multi_last_call_ms(self._h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
        return {"enqueue_ms": a.value, "total_ms": b.value, "rccl_allreduces": c.value}

    def close(self):
        if self._h:
            self._lib.gpsacq_multi_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def format_report(peaks, first_run=0):
    """SearchTask()'s per-run report (c/search_offline.cpp:264-287) for peaks of whole runs
    (32 consecutive tasks per run, reference schedule)."""
    out = []
    n_runs = len(peaks) // NUM_SATS
    for r in range(n_runs):
        pk = peaks[r * NUM_SATS:(r + 1) * NUM_SATS]
        hits = [sv for sv in range(NUM_SATS) if not (pk["snr"][sv] < THRESHOLD)]
        run = first_run + r
        out.append("%2d satellite: " % run + "".join("%5d " % sv for sv in hits) + "\n")
        out.append("%2d SNR(>=25): " % run + "".join("%5.1f " % pk["snr"][sv] for sv in hits) + "\n")
        out.append("%2d  lo_shift: " % run + "".join("%5d " % pk["lo_shift"][sv] for sv in hits) + "\n")
        out.append("%2d  ca_shift: " % run + "".join("%5d " % pk["ca_shift"][sv] for sv in hits) + "\n")
        out.append("".join("%2.0f " % pk["snr"][sv] for sv in range(NUM_SATS)) + "\n\n")
    return "".join(out)
