"""ctypes binding of oracle/liboracle_f{64,32}.so -- the CPU checker (test infrastructure)."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CELL_DTYPE = np.dtype([("max_pwr", "<f4"), ("max_i", "<i4"), ("tot_pwr", "<f4"), ("snr", "<f4")])
PEAK_DTYPE = np.dtype([("snr", "<f4"), ("lo_shift", "<i4"), ("ca_shift", "<i4"), ("max_pwr", "<f4")])
_libs = {}


def lib(kind="f64"):
    if kind in _libs:
        return _libs[kind]
    path = os.path.join(ROOT, "oracle", f"liboracle_{kind}.so")
    if not os.path.exists(path):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    L = ctypes.CDLL(path)
    vp, d, i = ctypes.c_void_p, ctypes.c_double, ctypes.c_int
    L.oracle_create.restype = vp
    L.oracle_create.argtypes = [d, d, d, i]
    L.oracle_destroy.argtypes = [vp]
    L.oracle_get_dmax.argtypes = [vp]
    L.oracle_get_nlags.argtypes = [vp]
    L.oracle_get_code_spectrum.argtypes = [vp, i, vp]
    L.oracle_sample.argtypes = [vp, vp]
    L.oracle_get_sample_spectrum.argtypes = [vp, vp]
    L.oracle_search_block.argtypes = [vp, vp, i, vp, vp]
    L.oracle_cell_power.argtypes = [vp, i, i, vp]
    L.oracle_sample_ramped.argtypes = [vp, vp, d]
    L.oracle_one_cell.argtypes = [vp, i, i, vp]
    L.oracle_search_file.argtypes = [vp, ctypes.c_char_p, i, ctypes.c_char_p, ctypes.c_size_t, vp, ctypes.c_size_t]
    L.oracle_bench_blocks.argtypes = [vp, vp, ctypes.c_long, ctypes.c_long, vp]
    L.oracle_bench_blocks.restype = ctypes.c_long
    L.oracle_code_replica.argtypes = [d, i, vp]
    L.oracle_lo_quadrants.argtypes = [d, d, i, vp]
    L.oracle_mix_block.argtypes = [vp, vp, vp]
    L.oracle_ca_chips.argtypes = [i, vp]
    L.oracle_search_code.argtypes = [i, i]
    L.oracle_dft.argtypes = [i, i, vp, vp]
    L.oracle_dmax.argtypes = [d, d]
    L.oracle_nlags.argtypes = [d]
    _libs[kind] = L
    return L


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


class Oracle:
    def __init__(self, fc, fs, max_fo=5000.0, ref_quirks=False, kind="f64"):
        self.L = lib(kind)
        self.fs = float(fs)
        self.max_fo = float(max_fo)
        self.h = self.L.oracle_create(fc, fs, max_fo, 1 if ref_quirks else 0)
        self.dmax = self.L.oracle_get_dmax(self.h)
        self.num_doppler = 2 * self.dmax + 1
        self.num_lags = self.L.oracle_get_nlags(self.h)

    def close(self):
        if self.h:
            self.L.oracle_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def code_spectrum(self, sv):
        out = np.zeros(80000, np.float32)
        self.L.oracle_get_code_spectrum(self.h, sv, _p(out))
        return out.view(np.complex64)

    def sample_spectrum(self, block):
        b = np.ascontiguousarray(np.frombuffer(block, dtype=np.uint8)[:5120])
        self.L.oracle_sample(self.h, _p(b))
        out = np.zeros(80000, np.float32)
        self.L.oracle_get_sample_spectrum(self.h, _p(out))
        return out.view(np.complex64)

    def search_block(self, block, sv):
        b = np.ascontiguousarray(np.frombuffer(block, dtype=np.uint8)[:5120])
        cells = np.zeros(self.num_doppler, CELL_DTYPE)
        peak = np.zeros(1, PEAK_DTYPE)
        self.L.oracle_search_block(self.h, _p(b), sv, _p(cells), _p(peak))
        return cells, peak[0]

    def search(self, bits, tasks=None):
        buf = np.frombuffer(bits, dtype=np.uint8)
        n_blocks = buf.size // 5120
        if tasks is None:
            tasks = [(b, b % 32) for b in range(n_blocks)]
        cells = np.zeros((len(tasks), self.num_doppler), CELL_DTYPE)
        peaks = np.zeros(len(tasks), PEAK_DTYPE)
        for t, (b, sv) in enumerate(tasks):
            c, p = self.search_block(buf[b * 5120:(b + 1) * 5120], sv)
            cells[t] = c
            peaks[t] = p
        return cells, peaks

    def search_grid(self, block, sv, sub=1, dstride=1, max_fo=None, points=None):
        """Restatement of the product's Doppler grid (gpsacq_set_doppler_step): step = bin * dstride / sub, points
        k = -K..K, K = trunc(max_fo / step); point k pairs the block's spectrum at sub-bin offset r / sub with the code
        spectrum shifted by d whole bins (k = d sub + r, or d = k dstride).  Returns (cells[points], ks)."""
        b = np.ascontiguousarray(np.frombuffer(block, dtype=np.uint8)[:5120])
        step = self.fs / 40000.0 * dstride / sub
        K = int((max_fo if max_fo is not None else self.max_fo) / step)
        ks = list(range(-K, K + 1)) if points is None else list(points)
        cells = np.zeros(len(ks), CELL_DTYPE)
        for r in range(sub):
            sel = [(i, k) for i, k in enumerate(ks) if (k % sub if sub > 1 else 0) == r]
            if not sel:
                continue
            self.L.oracle_sample_ramped(self.h, _p(b), float(r) / sub)
            one = np.zeros(1, CELL_DTYPE)
            for i, k in sel:
                d = (k - r) // sub if sub > 1 else k * dstride
                self.L.oracle_one_cell(self.h, sv, d, _p(one))
                cells[i] = one[0]
        return cells, ks

    def search_noncoherent(self, bits, stride, first_block, sv, n_acc, block_step, first_bin=None, n_bins=None, creep=False, align=False):
        """Restatement of the non-coherent extension: per Doppler bin, sum |IFFT|^2 per lag over
        n_acc blocks, then the reference's scan (:190-196) over the sum.  align: block k's powers are first moved back by
        k * ((block_step * stride * 8) mod S) samples (blocks that are not whole code periods apart).  creep: block k's powers are
        first moved back by round(k * c * bin) samples modulo the S lags, c = float32(samples between
        accumulated blocks * (fs / 40000) / L1) -- the product's code-creep compensation."""
        buf = np.frombuffer(bits, dtype=np.uint8)
        S = self.num_lags
        first_bin = -self.dmax if first_bin is None else first_bin
        n_bins = self.num_doppler if n_bins is None else n_bins
        power = np.zeros((n_bins, S), np.float32)
        tmp = np.zeros(S, np.float32)
        for k in range(n_acc):
            b = first_block + k * block_step
            blk = np.ascontiguousarray(buf[b * stride:b * stride + 5120])
            if blk.size < 5120:
                blk = np.concatenate([blk, np.zeros(5120 - blk.size, np.uint8)])
            self.L.oracle_sample(self.h, _p(blk))
            c = np.float32(float(block_step) * float(stride) * 8.0 * (self.fs / 40000.0) / 1575.42e6)
            for d in range(first_bin, first_bin + n_bins):
                self.L.oracle_cell_power(self.h, sv, d, _p(tmp))
                shift = int(np.rint(np.float32(k) * c * np.float32(d))) if creep else 0
                if align:  # gpsacq_set_block_alignment: the code phase between block starts that are not whole periods apart
                    shift += k * ((block_step * stride * 8) % S)
                power[d - first_bin] += np.roll(tmp, -shift)  # lag n -> (n - shift) mod S
        cells = np.zeros(n_bins, CELL_DTYPE)
        cells["max_pwr"] = power.max(axis=1)
        cells["max_i"] = power.argmax(axis=1)
        cells["tot_pwr"] = power.sum(axis=1, dtype=np.float64)
        cells["snr"] = cells["max_pwr"] / (cells["tot_pwr"] / S)
        return cells

    def search_file(self, path, max_runs=0):
        buf = ctypes.create_string_buffer(1 << 22)
        peaks = np.zeros(4096 * 32, PEAK_DTYPE)
        n = self.L.oracle_search_file(self.h, path.encode(), max_runs, buf, len(buf), _p(peaks), peaks.size)
        return n, buf.value.decode(), peaks[:max(n, 0) * 32]

    def bench_blocks(self, bits, n_blocks):
        buf = np.ascontiguousarray(np.frombuffer(bits, dtype=np.uint8))
        peaks = np.zeros(n_blocks, PEAK_DTYPE)
        cells = self.L.oracle_bench_blocks(self.h, _p(buf), n_blocks, 5120, _p(peaks))
        return cells, peaks
