"""iq8_oracle.py -- CPU restatement (numpy, float64) of the reference's MATLAB pre-processing
that turns an 8-bit IQ capture into gps_test's 1-bit input.  TEST INFRASTRUCTURE ONLY.

Restates, operation by operation:
  proc_rtl_bin_for_gps.m:12-26   uint8 -> y-128 -> I + 1i*Q -> y - mean(y) -> real(y) -> (1-sign)/2 -> ubit1
  proc_rtl_bin_for_gps.m:31-47   same, with y = real(y.' .* exp(1i.*2.*pi.*fc.*(0:n-1).*(1./fs)))
  proc_hackrf_bin_for_gps.m:7-19 int8 input, no offset; its mixer phase is written 1i.*(0:n-1).*fc.*2.*pi./fs
MATLAB evaluates these left to right in double: theta = (((2*pi)*fc)*n)*(1/fs) for the rtl script, (((n*fc)*2)*pi)/fs for the
HackRF one (mixer_phase below; the two differ by an ulp of theta now and then).
fwrite(..., 'ubit1') rounds the value 0.5 that (1-sign(0))/2 produces to 1 and packs LSB first
(the bit order Sample() unpacks, c/search_offline.cpp:143-146).
PIN STATUS: unpinned -- neither MATLAB nor Octave exists in this image and the reference ships no
converted file; this is a restatement of the script text only.
"""
import numpy as np


def mixer_phase(n, fc, fs, signed):
    """The mixer phase as the script of the format evaluates it, left to right in double:
    proc_rtl_bin_for_gps.m:41     exp(1i.*2.*pi.*fc.*(0:n-1).*(1./fs))   -> (((2 pi) fc) n) (1 / fs)
    proc_hackrf_bin_for_gps.m:14  exp(1i.*(0:n-1).*2.6e6.*2.*pi./10e6)    -> (((n fc) 2) pi) / fs"""
    if signed:
        return (((n * fc) * 2.0) * np.pi) / fs
    return (((2.0 * np.pi) * fc) * n) * (1.0 / fs)


def iq8_to_bits(raw, signed=False, remove_dc=True, mix_hz=0.0, fs=2.8e6):
    raw = np.asarray(raw).view(np.uint8).ravel()
    y = raw.view(np.int8).astype(np.float64) if signed else raw.astype(np.float64) - 128.0
    y = y[0::2] + 1j * y[1::2]
    if remove_dc:
        y = y - np.mean(y)
    if mix_hz != 0.0:
        n = np.arange(y.size, dtype=np.float64)
        theta = mixer_phase(n, mix_hz, fs, signed)
        r = y.real * np.cos(theta) - y.imag * np.sin(theta)
    else:
        r = y.real
    bits = np.where(r > 0, 0, 1).astype(np.uint8)
    return np.packbits(bits, bitorder="little")


def iq8_to_real(raw, signed=False, remove_dc=True, mix_hz=0.0, fs=2.8e6):
    """The same value before the sign: what the product's multi-bit path (gpsacq_iq8_input.multibit; SURVEY.md section 8f.1
    "direct float path", no reference counterpart) feeds the forward transform, as float32."""
    raw = np.asarray(raw).view(np.uint8).ravel()
    y = raw.view(np.int8).astype(np.float64) if signed else raw.astype(np.float64) - 128.0
    y = y[0::2] + 1j * y[1::2]
    if remove_dc:
        y = y - np.mean(y)
    if mix_hz != 0.0:
        n = np.arange(y.size, dtype=np.float64)
        theta = mixer_phase(n, mix_hz, fs, signed)
        r = y.real * np.cos(theta) - y.imag * np.sin(theta)
    else:
        r = y.real
    return r.astype(np.float32)


def multibit_cells(r_block, lo_quadrant, code_replica, dmax, n_lags, eps=0.0, dops=None):
    """Correlate() (c/search_offline.cpp:169-201) on a block of multi-bit samples, float64 numpy: the LO of Sample() (:143-153)
    applied as signs (lo_cos = {0,1,1,0}, lo_sin = {1,1,0,0}; 1 <-> factor -1), then exactly the reference's search.
    r_block: >= 40000 float samples; lo_quadrant: int(lo_phase) per sample; code_replica: SearchInit's 40000 floats.
    Returns (max_pwr, max_i, tot_pwr) arrays over dop = -dmax..dmax."""
    N = 40000
    lo_sin = np.array([1, 1, 0, 0])
    lo_cos = np.array([0, 1, 1, 0])
    x = r_block[:N].astype(np.float64)
    x = x * (1.0 - 2.0 * lo_cos[lo_quadrant[:N]]) + 1j * x * (1.0 - 2.0 * lo_sin[lo_quadrant[:N]])
    return complex_cells(x, code_replica, dmax, n_lags, eps, dops)


def multibit_pwr(r_block, lo_quadrant, code_replica, dmax, n_lags):
    """The per-lag powers behind multibit_cells, [2 dmax + 1][n_lags] (what the non-coherent mode sums over blocks)."""
    N = 40000
    lo_sin = np.array([1, 1, 0, 0])
    lo_cos = np.array([0, 1, 1, 0])
    x = r_block[:N].astype(np.float64)
    x = x * (1.0 - 2.0 * lo_cos[lo_quadrant[:N]]) + 1j * x * (1.0 - 2.0 * lo_sin[lo_quadrant[:N]])
    D = np.fft.fft(x)
    C = np.fft.fft(np.asarray(code_replica, dtype=np.float64))
    out = np.empty((2 * dmax + 1, n_lags))
    for d in range(-dmax, dmax + 1):
        y = np.fft.ifft(np.conj(D) * np.roll(C, d)) * N
        out[d + dmax] = y[:n_lags].real ** 2 + y[:n_lags].imag ** 2
    return out


def iq8_to_complex(raw, signed=False, remove_dc=True, mix_hz=0.0, fs=2.8e6):
    """The complex sample itself (gpsacq_iq8_input.multibit = 2: the capture is at baseband already): (y - mean(y)) turned by
    exp(i theta), theta as in iq8_to_bits; complex64 like the device buffer."""
    raw = np.asarray(raw).view(np.uint8).ravel()
    y = raw.view(np.int8).astype(np.float64) if signed else raw.astype(np.float64) - 128.0
    y = y[0::2] + 1j * y[1::2]
    if remove_dc:
        y = y - np.mean(y)
    if mix_hz != 0.0:
        n = np.arange(y.size, dtype=np.float64)
        theta = mixer_phase(n, mix_hz, fs, signed)
        cs, sn = np.cos(theta), np.sin(theta)
        y = (y.real * cs - y.imag * sn) + 1j * (y.real * sn + y.imag * cs)
    return y.astype(np.complex64)


def hackrf_replay_file(bits, lo_quadrant):
    """c/conv_1bit_bin_to_hackrf_bin.cpp:18-20,30-31,61-80 restated: every 1-bit sample (LSB first) XOR-mixed with the converter's
    OWN quadrature tables lo_sin = {1,1,0,0}, lo_cos = {1,0,0,1} (not Sample()'s: there lo_cos = {0,1,1,0} and the components are
    swapped, search_offline.cpp:124-125,149-150 -- together a turn of the whole stream by +j, which no power notices), Bipolar = -30
    for 1, +30 for 0, written I then Q as int8.  lo_quadrant: int(lo_phase) per sample of the WHOLE stream (the converter's NCO
    runs on across blocks; oracle_lo_quadrants() with n = all samples).  The converter only handles whole 55.8 MB reads (:26,55-58);
    this is its inner loop on a stream of any length."""
    lo_sin = np.array([1, 1, 0, 0], dtype=np.uint8)
    lo_cos = np.array([1, 0, 0, 1], dtype=np.uint8)
    b = np.unpackbits(np.asarray(bits, dtype=np.uint8), bitorder="little")
    q = np.asarray(lo_quadrant)[:b.size]
    out = np.empty(2 * b.size, dtype=np.int8)
    out[0::2] = np.where(b ^ lo_sin[q], -30, 30)
    out[1::2] = np.where(b ^ lo_cos[q], -30, 30)
    return out


def hackrf_baseband_file_matlab(bits):
    """gps_bin1bit_log2bin.m:15-32 restated: the 1-bit capture (ubit1, LSB first) as +-1, times the fs/4 LO lo_real = [1 0 -1 0 ...],
    lo_imag = [0 1 0 -1 ...] running from the start of the file, times 100, interleaved I,Q as int8 --
    `gps.samples.8bit.IQ.fs5456.baseband.bin`, the Nottingham capture made ready for HackRF replay.  For IF = 3/4 fs this is Sample()'s
    XOR mixer (c/search_offline.cpp:143-153: quadrants 0,3,2,1 -> (1-j, 1+j, -1+j, -1-j) s) up to the constant 100 / (1-j)."""
    b = np.unpackbits(np.asarray(bits, dtype=np.uint8), bitorder="little").astype(np.int16)
    y = 1 - 2 * b
    n = np.arange(y.size) % 4
    lo_real = np.array([1, 0, -1, 0], dtype=np.int16)[n]
    lo_imag = np.array([0, 1, 0, -1], dtype=np.int16)[n]
    out = np.empty(2 * y.size, dtype=np.int8)
    out[0::2] = y * lo_real * 100
    out[1::2] = y * lo_imag * 100
    return out


def complex_cells(x, code_replica, dmax, n_lags, eps=0.0, dops=None):
    """Correlate() (c/search_offline.cpp:169-201) on 40000 complex samples (what Sample() would have left in fwd_buf), float64.
    eps: the samples (as floats) turned by exp(-2 pi i eps n / N) first, the sub-bin carrier offset of
    gpsacq_oracle.c::oracle_sample_ramped (product rounded to float); dops: the whole-bin shifts to evaluate (default -dmax..dmax)."""
    N = 40000
    x = np.asarray(x[:N], dtype=np.complex64).astype(np.complex128)
    if eps != 0.0:
        x = (x * np.exp(-2j * np.pi * eps * np.arange(N) / N)).astype(np.complex64).astype(np.complex128)
    D = np.fft.fft(x)
    C = np.fft.fft(np.asarray(code_replica, dtype=np.float64))
    mp, mi, tp = [], [], []
    for d in (range(-dmax, dmax + 1) if dops is None else dops):
        y = np.fft.ifft(np.conj(D) * np.roll(C, d)) * N
        pwr = y[:n_lags].real ** 2 + y[:n_lags].imag ** 2
        mp.append(pwr.max())
        mi.append(int(pwr.argmax()))
        tp.append(pwr.sum())
    return np.array(mp), np.array(mi), np.array(tp)

