"""gps_sig_gen.m (the reference's test-signal generator) restated: the oracle reproduces the reference's own bundled
output file bit for bit; the device generator (gpsacq_generate_sig) is checked against the same file on the GPU."""
import hashlib
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def _chips(prn):
    from oracle_lib import lib, _p
    c = np.zeros(1023, np.uint8)
    lib("f64").oracle_ca_chips(prn - 1, _p(c))
    return c


def test_oracle_reproduces_the_reference_file(golden_dir):
    import sig_gen_oracle as sg
    ref = open(os.path.join(golden_dir, "gps_sig_tmp.bin"), "rb").read()
    assert hashlib.sha256(ref).hexdigest().startswith("a6242849")  # the reference's file
    db = json.load(open(os.path.join(golden_dir, "gps_sig_tmp_databits.json")))
    out = sg.generate(_chips(db["prn"]), db["bits_pm1"])
    assert len(out) == len(ref) == 2046006
    assert out.tobytes() == ref
    # the navigation bits can be read back out of a generated capture, every chip agreeing
    data, worst = sg.recover_data_bits(out.tobytes(), _chips(db["prn"]))
    assert worst == 1.0 and [int(v) for v in data] == db["bits_pm1"]


def test_raised_cosine_taps():
    import sig_gen_oracle as sg
    h = sg.rcosine_taps()
    assert len(h) == 49 and h[24] == 1.0 and np.allclose(h, h[::-1], rtol=0, atol=0)
    assert abs(h[20] - 0.6002108774380708) < 1e-15           # t = 0.5
    assert all(abs(h[24 + 8 * k]) < 1e-16 for k in (-3, -2, -1, 1, 2, 3))  # Nyquist zeros (to rounding)
    assert h[16] == 0.25 * np.sin(np.pi)                      # the singular point t = -1: (R/2) sin(pi/(2R))


def test_product_tap_table_is_rcosine(golden_dir):
    """the 49 doubles embedded in gen_kernels.hip are rcosine(1, 8) as the oracle computes it, bit for bit"""
    import re
    import sig_gen_oracle as sg
    src = open(os.path.join(ROOT, "gnss-gps-sdr_amd", "csrc", "gen_kernels.hip")).read()
    body = src[src.index("__constant__ double c_rc[49] = {"):]
    body = body[:body.index("};")]
    vals = [float.fromhex(v) for v in re.findall(r"-?0x1\.[0-9a-f]+p[+-]\d+", body)]
    assert len(vals) == 49 and np.array_equal(np.array(vals), sg.rcosine_taps())


# ---- gps_sig_gen.m:21-30: the HackRF transmit file (second reference-defined known answer, complex-baseband format) ----
def test_tx_restatement_against_the_reference_held_1bit_file(golden_dir):
    """hackrf_tx() is the same shaped baseband as the bundled 1-bit file's, x 50, at IF 0.  Pins: (1) on every even sample the
    fs/4 carrier of the 1-bit leg is exactly +-1, so the sign of the transmit file's I there must BE the reference file's bit
    (wherever int8 did not round it to 0) -- the reference's own file checks the restated baseband of the first repetition;
    (2) chip centres read exactly +-50 x data x chip, Q is zero, the stream has 5 x 16 368 000 + 48 samples; (3) the written
    int8 does not depend on the order conv() adds its terms (undocumented for conv(short, long)): both orders, every sample."""
    import sig_gen_oracle as sg
    db = json.load(open(os.path.join(golden_dir, "gps_sig_tmp_databits.json")))
    chips = _chips(db["prn"])
    ref_bits = np.unpackbits(np.fromfile(os.path.join(golden_dir, "gps_sig_tmp.bin"), dtype=np.uint8), bitorder="little")
    assert sg.tx_samples(len(db["bits_pm1"])) == 5 * 16368000 + 48
    per_rep = 16368000
    for first, count in ((0, 3000000), (per_rep - 1000000, 1000000)):
        iq = sg.hackrf_tx(chips, db["bits_pm1"], first=first, count=count)
        assert np.array_equal(iq, sg.hackrf_tx(chips, db["bits_pm1"], first=first, count=count, newest_first=True))
        i, q = iq[0::2].astype(np.int32), iq[1::2]
        assert not q.any() and np.abs(i).max() <= 75
        m = np.arange(first, first + count)
        bit = ref_bits[m]
        e0, e2 = (m % 4 == 0) & (i != 0), (m % 4 == 2) & (i != 0)
        assert e0.sum() > 0.2 * count and e2.sum() > 0.2 * count  # (I rounds to 0 on ~7 % of the samples: chip transitions)
        assert np.array_equal(bit[e0], (i[e0] < 0).astype(np.uint8))   # carrier +1: bit = (s < 0)
        assert np.array_equal(bit[e2], (i[e2] > 0).astype(np.uint8))   # carrier -1: bit = (s > 0)
        centre = (m % 8 == 0) & (m >= 24)
        k = (m[centre] - 24) // 8
        g = 1 - 2 * chips.astype(np.int32)
        d = np.asarray(db["bits_pm1"])[(k // 20460) % 100]
        assert np.array_equal(i[centre], 50 * d * g[k % 1023])
    # the second repetition repeats the first (16 368 000 samples = 2000 whole code periods, 100 whole bits); the tail is the filter's
    a = sg.hackrf_tx(chips, db["bits_pm1"], first=48, count=200000)
    b = sg.hackrf_tx(chips, db["bits_pm1"], first=per_rep + 48, count=200000)
    assert np.array_equal(a, b)
    tail = sg.hackrf_tx(chips, db["bits_pm1"], first=5 * per_rep, count=48)
    assert tail[0::2].any() and not tail[1::2].any()


def test_tx_file_searched_as_complex_baseband_on_the_cpu(golden_dir):
    """The known answer the script defines for its transmit file, through Correlate() restated on complex samples (float64):
    PRN 8 at Doppler bin 0, code phase (40960 b - 20) mod 8184 at block b -- the same law as the 1-bit file's
    (test_oracle.py::test_code_phase_follows_from_gps_sig_gen), also across the seam of two repetitions (block 400)."""
    import sig_gen_oracle as sg
    from iq8_oracle import complex_cells
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from make_golden import code_replica
    db = json.load(open(os.path.join(golden_dir, "gps_sig_tmp_databits.json")))
    chips, fs = _chips(db["prn"]), 8.184e6
    for b in (0, 7, 399, 400):
        iq = sg.hackrf_tx(chips, db["bits_pm1"], first=40960 * b, count=40000)
        z = iq[0::2].astype(np.float64) + 1j * iq[1::2].astype(np.float64)
        mp, mi, tp = complex_cells(z, code_replica(fs, db["prn"] - 1), 24, 8184)
        snr = mp / (tp / 8184)
        k = int(np.argmax(snr))
        # block 399 holds the seam 24 960 samples in, where navigation bit 99 (+1) meets bit 0 (-1): the two parts of the block
        # pull against each other (a quarter of the amplitude is left) and the broad peak (8 samples per chip) tips to the next lag
        want = (40960 * b - 20) % 8184
        assert k - 24 == 0 and snr[k] > 300 and (mi[k] == want if b != 399 else mi[k] == want - 1)
        mp2, _, tp2 = complex_cells(z, code_replica(fs, 3), 24, 8184)  # another PRN: nothing
        assert (mp2 / (tp2 / 8184)).max() < 40


@pytest.mark.gpu
def test_device_generator_reproduces_the_reference_file(golden_dir):
    """gpsacq_generate_sig (gen_kernels.hip, k_siggen) against the reference's own gps_sig_tmp.bin: every one of the
    16 368 048 samples."""
    import gpsacq
    ref = np.frombuffer(open(os.path.join(golden_dir, "gps_sig_tmp.bin"), "rb").read(), dtype=np.uint8)
    db = json.load(open(os.path.join(golden_dir, "gps_sig_tmp_databits.json")))
    with gpsacq.Engine(2.046e6, 8.184e6, 5000.0) as eng:
        out = eng.generate_sig(db["prn"], db["bits_pm1"])
        assert out.size == ref.size
        diff = np.unpackbits(out ^ ref).sum()
        assert diff == 0, f"{diff} of {8 * ref.size} samples differ"
        # another PRN / other bits: found by the search at zero Doppler like the reference's file
        out2 = eng.generate_sig(21, [1, -1, -1, 1])  # 81 846 bytes = 15 whole blocks
        _, pk = eng.search(out2, tasks=[(0, sv) for sv in range(32)] + [(9, 20)])
        assert pk["snr"][20] > 300 and pk["lo_shift"][20] == 0 and int(np.argmax(pk["snr"][:32])) == 20
        assert pk["snr"][32] > 300 and pk["lo_shift"][32] == 0
