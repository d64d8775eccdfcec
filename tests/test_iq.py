"""8-bit IQ ingestion: the numpy oracle on hand-checkable cases (CPU) and
the device kernels against it (GPU)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from iq8_oracle import iq8_to_bits  # noqa: E402


def test_oracle_small_cases():
    # rtl format: I = {130,126,128,128}, Q irrelevant without mixing; mean(I-128) = 0 -> r = {2,-2,0,0}
    raw = np.array([130, 9, 126, 9, 128, 9, 128, 9], dtype=np.uint8)
    assert iq8_to_bits(raw, remove_dc=True)[0] == 0b1110  # r>0 -> 0; r<0 -> 1; r==0 -> 1 (fwrite rounds 0.5 up)
    assert iq8_to_bits(raw, remove_dc=False)[0] == 0b1110
    # DC removal flips what an offset hides: I-128 = {5,3,5,3} -> mean 4 -> {+1,-1,+1,-1}
    raw = np.array([133, 0, 131, 0, 133, 0, 131, 0], dtype=np.uint8)
    assert iq8_to_bits(raw, remove_dc=False)[0] == 0b0000 and iq8_to_bits(raw, remove_dc=True)[0] == 0b1010
    # HackRF int8, mixing by fs/4: real((I + jQ) e^{j pi n / 2}) = I, -Q, -I, Q, ...
    iq = np.array([10, 3, 10, 3, 10, 3, 10, 3], dtype=np.int8)
    assert iq8_to_bits(iq, signed=True, remove_dc=False, mix_hz=0.25, fs=1.0)[0] == 0b0110
    # packing: sample n -> bit n % 8 of byte n // 8, tail bits zero
    raw = np.zeros(2 * 11, dtype=np.uint8) + 128
    raw[2 * 9] = 200
    out = iq8_to_bits(raw, remove_dc=False)
    assert out.size == 2 and out[0] == 0xFF and out[1] == 0b101


def test_committed_bits_fixture_reproducible(golden_dir):
    raw = np.fromfile(os.path.join(golden_dir, "synth_iq8_rtl.bin"), dtype=np.uint8)
    want = np.fromfile(os.path.join(golden_dir, "synth_iq8_rtl_bits.bin"), dtype=np.uint8)
    assert np.array_equal(iq8_to_bits(raw, remove_dc=True, mix_hz=0.62e6, fs=2.8e6), want)


def test_multibit_restatement_reduces_to_the_1bit_oracle(golden_dir):
    """oracle/iq8_oracle.py::multibit_cells (float samples, LO applied as signs) on samples that are only their sign must
    give the cells of the C oracle's Sample() + Correlate() on the same 1-bit block: pins the sign convention of the
    multi-bit path to the reference's XOR mixer (c/search_offline.cpp:143-153)."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from iq8_oracle import multibit_cells
    from make_golden import lo_quadrants, code_replica
    from oracle_lib import Oracle
    fc, fs = 0.62e6, 2.8e6
    buf = open(os.path.join(golden_dir, "synth_rtl_fs2800.bin"), "rb").read()
    blk = np.frombuffer(buf[7 * 5120:8 * 5120], np.uint8)
    orc = Oracle(fc, fs, 5000.0)
    cells, _ = orc.search_block(blk, 7)
    r = (1.0 - 2.0 * np.unpackbits(blk, bitorder="little").astype(np.float64)).astype(np.float32)
    mp, mi, tp = multibit_cells(r, lo_quadrants(fc, fs, 40960), code_replica(fs, 7), orc.dmax, orc.num_lags)
    np.testing.assert_allclose(mp, cells["max_pwr"], rtol=2e-6)
    np.testing.assert_allclose(tp, cells["tot_pwr"], rtol=1e-5)
    assert np.array_equal(mi, cells["max_i"])


def _quadrants(fc, fs, n):
    from oracle_lib import lib, _p
    q = np.zeros(n, np.uint8)
    lib().oracle_lo_quadrants(fc, fs, n, _p(q))
    return q


def test_converter_restatement_is_samples_fwd_buf_turned_by_j(golden_dir):
    """oracle/iq8_oracle.py::hackrf_replay_file (c/conv_1bit_bin_to_hackrf_bin.cpp's inner loop) on the bundled capture: its
    I + jQ is 30 x the C oracle's Sample() buffer (c/search_offline.cpp:143-153) divided by j in every block -- the converter's
    own LO tables and component order differ from Sample()'s by exactly that turn -- so Correlate() on it gives 900 x the powers
    of the 1-bit search, the same max_i and the same SNR.  fs/4 IF: the NCO step is 1.0, so running on across blocks (converter)
    and restarting per block (Sample) are the same LO."""
    from iq8_oracle import hackrf_replay_file, complex_cells
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from make_golden import code_replica
    from oracle_lib import Oracle, lib, _p
    fs = 8.184e6
    fc = fs / 4
    bits = np.fromfile(os.path.join(golden_dir, "gps_sig_tmp.bin"), dtype=np.uint8)[:9 * 5120]
    iq = hackrf_replay_file(bits, _quadrants(fc, fs, bits.size * 8))
    assert set(np.unique(iq)) == {-30, 30}
    z = iq[0::2].astype(np.float64) + 1j * iq[1::2].astype(np.float64)
    quad = _quadrants(fc, fs, 40960)
    orc = Oracle(fc, fs, 5000.0)
    for b, sv in ((0, 0), (7, 7), (8, 7)):
        blk = np.ascontiguousarray(bits[b * 5120:(b + 1) * 5120])
        mixed = np.zeros(2 * 40960, np.float32)
        lib().oracle_mix_block(_p(blk), _p(quad), _p(mixed))
        want = mixed.view(np.complex64).astype(np.complex128)
        assert np.array_equal(1j * z[b * 40960:(b + 1) * 40960], 30.0 * want)
        cells, peak = orc.search_block(blk, sv)
        mp, mi, tp = complex_cells(z[b * 40960:], code_replica(fs, sv), orc.dmax, orc.num_lags)
        np.testing.assert_allclose(mp / 900.0, cells["max_pwr"], rtol=2e-6)
        np.testing.assert_allclose(tp / 900.0, cells["tot_pwr"], rtol=1e-5)
        assert np.array_equal(mi, cells["max_i"])


def test_matlab_converter_restatement_is_samples_fwd_buf_up_to_a_constant(golden_dir):
    """gps_bin1bit_log2bin.m's int8 baseband file of a fs 5.456 MHz / IF 4.092 MHz capture: (1 - j) x its I + jQ = 100 x the C
    oracle's Sample() buffer, sample for sample."""
    from iq8_oracle import hackrf_baseband_file_matlab
    from oracle_lib import lib, _p
    bits = np.fromfile(os.path.join(golden_dir, "synth_nott_fs5456.bin"), dtype=np.uint8)[:3 * 5120]
    iq = hackrf_baseband_file_matlab(bits)
    z = iq[0::2].astype(np.float64) + 1j * iq[1::2].astype(np.float64)
    quad = _quadrants(4.092e6, 5.456e6, 40960)
    for b in range(3):
        blk = np.ascontiguousarray(bits[b * 5120:(b + 1) * 5120])
        mixed = np.zeros(2 * 40960, np.float32)
        lib().oracle_mix_block(_p(blk), _p(quad), _p(mixed))
        assert np.array_equal(z[b * 40960:(b + 1) * 40960] * (1 - 1j), 100.0 * mixed.view(np.complex64).astype(np.complex128))


@pytest.mark.gpu
def test_complex_baseband_search_of_the_converters_file_equals_the_1bit_search(golden_dir):
    """gpsacq_iq8_input.multibit = 2 (GPSACQ_SAMPLES_COMPLEX) on the file the reference's own converter would write from the
    bundled capture (all 12 runs, 31 MB of int8 IQ): every cell's power is 900 x the 1-bit search's, every reported peak the
    same.  Also the Nottingham-rate fixture (IF = 3/4 fs: NCO step 3.0, again exact) and, with a residual IF, the cells against
    the float64 restatement."""
    import gpsacq
    from iq8_oracle import hackrf_replay_file, iq8_to_complex, complex_cells
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from make_golden import code_replica
    for name, fc, fs in (("gps_sig_tmp.bin", 2.046e6, 8.184e6), ("synth_nott_fs5456.bin", 4.092e6, 5.456e6)):
        bits = np.fromfile(os.path.join(golden_dir, name), dtype=np.uint8)
        n_blocks = bits.size // 5120
        bits = bits[:n_blocks * 5120]
        iq = hackrf_replay_file(bits, _quadrants(fc, fs, bits.size * 8))
        with gpsacq.Engine(fc, fs, 5000.0) as eng:
            c1, p1 = eng.search(bits)
            inp = eng.iq8_input(signed=True, remove_dc=False, total_samples=iq.size // 2, multibit=2)
            c2, p2 = eng.search_iq8(iq, inp)
        assert c1.shape == c2.shape and c1.shape[0] == n_blocks
        np.testing.assert_allclose(c2["max_pwr"] / 900.0, c1["max_pwr"], rtol=3e-6)
        np.testing.assert_allclose(c2["tot_pwr"] / 900.0, c1["tot_pwr"], rtol=3e-6)
        np.testing.assert_allclose(c2["snr"], c1["snr"], rtol=5e-6)
        # the argmax of a cell can only move where two lags tie to rounding
        diff = c1["max_i"] != c2["max_i"]
        assert diff.mean() < 1e-3, diff.sum()
        if diff.any():
            np.testing.assert_allclose(c2["max_pwr"][diff] / 900.0, c1["max_pwr"][diff], rtol=3e-6)
        hit = p1["snr"] >= 25
        assert hit.any()
        assert np.array_equal(p1["lo_shift"][hit], p2["lo_shift"][hit]) and np.array_equal(p1["ca_shift"][hit], p2["ca_shift"][hit])
        np.testing.assert_allclose(p2["snr"], p1["snr"], rtol=5e-6)
    # the MATLAB converter of the Nottingham capture (gps_bin1bit_log2bin.m: +-100 and 0, fs/4 LO): for IF = 3/4 fs the same
    # baseband stream up to the constant 100 / (1 - j), so every power is 100^2 / 2 = 5000 x the 1-bit search's
    from iq8_oracle import hackrf_baseband_file_matlab
    fc, fs = 4.092e6, 5.456e6
    bits = np.fromfile(os.path.join(golden_dir, "synth_nott_fs5456.bin"), dtype=np.uint8)
    bits = bits[:(bits.size // 5120) * 5120]
    iqm = hackrf_baseband_file_matlab(bits)
    assert set(np.unique(iqm)) == {-100, 0, 100}
    with gpsacq.Engine(fc, fs, 5000.0) as eng:
        c1, p1 = eng.search(bits)
        c2, p2 = eng.search_iq8(iqm, eng.iq8_input(signed=True, remove_dc=False, total_samples=iqm.size // 2, multibit=2))
    np.testing.assert_allclose(c2["max_pwr"] / 5000.0, c1["max_pwr"], rtol=3e-6)
    np.testing.assert_allclose(c2["tot_pwr"] / 5000.0, c1["tot_pwr"], rtol=3e-6)
    assert (c1["max_i"] != c2["max_i"]).mean() < 1e-3
    hit = p1["snr"] >= 25
    assert hit.sum() >= 5 and np.array_equal(p1["lo_shift"][hit], p2["lo_shift"][hit]) and np.array_equal(p1["ca_shift"][hit], p2["ca_shift"][hit])
    # a residual IF: the converter's file turned down by 700 Hz is searched with mix_hz = +700 (and the mean removed, to cover it)
    fc, fs = 2.046e6, 8.184e6
    bits = np.fromfile(os.path.join(golden_dir, "gps_sig_tmp.bin"), dtype=np.uint8)[7 * 5120:9 * 5120]
    iq = hackrf_replay_file(bits, _quadrants(fc, fs, bits.size * 8))
    z = (iq[0::2] + 1j * iq[1::2]) * np.exp(-2j * np.pi * 700.0 * np.arange(iq.size // 2) / fs) * 4.0 + (3.0 - 2.0j)
    iq2 = np.empty(iq.size, np.int8)
    iq2[0::2] = np.clip(np.round(z.real), -128, 127)
    iq2[1::2] = np.clip(np.round(z.imag), -128, 127)
    with gpsacq.Engine(fc, fs, 5000.0) as eng:
        mean = eng.iq8_mean(iq2, signed=True)
        inp = eng.iq8_input(signed=True, remove_dc=True, mean=mean, mix_hz=700.0, fs=fs, total_samples=iq2.size // 2, multibit=2)
        cells, peaks = eng.search_iq8(iq2, inp, tasks=[(0, 7), (1, 7)])
        _, p1 = eng.search(bits, tasks=[(0, 7), (1, 7)])
        zz = iq8_to_complex(iq2, signed=True, remove_dc=True, mix_hz=700.0, fs=fs)
        for t in range(2):
            mp, mi, tp = complex_cells(zz[t * 40960:], code_replica(fs, 7), eng.dmax, eng.num_lags)
            np.testing.assert_allclose(cells["max_pwr"][t], mp, rtol=2e-5)
            np.testing.assert_allclose(cells["tot_pwr"][t], tp, rtol=2e-5)
            assert (cells["max_i"][t] != mi).sum() <= 1
            assert peaks["lo_shift"][t] == p1["lo_shift"][t] and peaks["ca_shift"][t] == p1["ca_shift"][t] and peaks["snr"][t] > 25
        with pytest.raises(gpsacq.GpsAcqError):
            eng.search_iq8(iq2, eng.iq8_input(signed=True, multibit=3))
        # the complex path on a Doppler grid finer than a bin: again 900 x the 1-bit search of the same grid (whose sub-bin turn is
        # in the transform's twiddles; here it is applied to the samples)
        eng.set_doppler_step(80.0)
        assert eng.doppler_sub == 3
        inp = eng.iq8_input(signed=True, remove_dc=False, total_samples=iq.size // 2, multibit=2)
        cg, pg = eng.search_iq8(iq, inp, tasks=[(0, 7), (1, 3)])
        cb, pb = eng.search(bits, tasks=[(0, 7), (1, 3)])
        np.testing.assert_allclose(cg["max_pwr"] / 900.0, cb["max_pwr"], rtol=5e-6)
        np.testing.assert_allclose(cg["tot_pwr"] / 900.0, cb["tot_pwr"], rtol=5e-6)
        assert (cg["max_i"] != cb["max_i"]).sum() <= 1
        assert pg["lo_shift"][0] == pb["lo_shift"][0] and pg["ca_shift"][0] == pb["ca_shift"][0]


@pytest.mark.gpu
def test_device_conversion_matches_oracle(golden_dir):
    import gpsacq
    raw = np.fromfile(os.path.join(golden_dir, "synth_iq8_rtl.bin"), dtype=np.uint8)
    want = np.fromfile(os.path.join(golden_dir, "synth_iq8_rtl_bits.bin"), dtype=np.uint8)
    rng = np.random.default_rng(3)
    with gpsacq.Engine(0.62e6, 2.8e6, 5000.0) as eng:
        got = eng.iq8_to_bits(raw, remove_dc=True, mix_hz=0.62e6, fs=2.8e6)
        assert np.array_equal(got, want)
        # other modes and ragged lengths, bit-exact against the oracle
        for n, signed, dc, mix in [(8, False, True, 0.0), (13, True, False, 0.7e6), (4099, False, True, 0.62e6),
                                   (100003, True, True, 1.0e6), (81920, False, False, 0.0)]:
            r = rng.integers(0, 256, 2 * n, dtype=np.uint8)
            a = eng.iq8_to_bits(r, signed=signed, remove_dc=dc, mix_hz=mix, fs=2.8e6)
            b = iq8_to_bits(r, signed=signed, remove_dc=dc, mix_hz=mix, fs=2.8e6)
            assert a.size == b.size
            assert np.unpackbits(a ^ b).sum() <= max(0, n // 1000000), (n, signed, dc, mix)
        # end to end: the converted capture is a valid gps_test input and PRN 5 is found where injected
        cells, peaks = eng.search(got.tobytes(), tasks=[(0, sv) for sv in range(32)], stride=5120)
        best = int(np.argmax(peaks["snr"]))
        assert best == 4 and peaks["snr"][4] > 25
        assert abs(int(peaks["lo_shift"][4]) - round(1023.0 * 40000 / 2.8e6)) <= 1
