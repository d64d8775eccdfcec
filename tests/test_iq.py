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
