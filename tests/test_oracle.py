"""CPU tests of the oracle (the C restatement under oracle/): pinned against the reference-held capture gps_sig_tmp.bin through
the parameters of the script that made it (gps_sig_gen.m: PRN 8, Doppler 0, the code phase of every block), the README's known
answer, IS-GPS-200 chips, and an independent float64 numpy restatement (tests/golden/make_golden.py).  The transcript recorded
in BASELINE.md section 2 (tests/golden/ref_known_answers.json) is the survey's MKL-shim build of the reference sources -- a
stand-in build that pins nothing by itself -- and is compared as a regression transcript only."""
import json
import sys
import os

import numpy as np
import pytest

from oracle_lib import Oracle, lib, _p

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_dft_matches_numpy():
    L = lib("f64")
    rng = np.random.default_rng(5)
    for n in (5, 20, 100, 5000, 40000):
        x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
        for direction, ref in ((-1, np.fft.fft(x.astype(np.complex128))), (+1, np.fft.ifft(x.astype(np.complex128)) * n)):
            out = np.zeros(n, np.complex64)
            assert L.oracle_dft(n, direction, _p(x), _p(out)) == 0
            assert np.abs(out - ref).max() / np.abs(ref).max() < 2e-7
    Lf = lib("f32")
    x = (rng.standard_normal(40000) + 1j * rng.standard_normal(40000)).astype(np.complex64)
    out = np.zeros(40000, np.complex64)
    Lf.oracle_dft(40000, -1, _p(x), _p(out))
    ref = np.fft.fft(x.astype(np.complex128))
    assert np.abs(out - ref).max() / np.abs(ref).max() < 5e-6


def test_ca_code_properties():
    """c/cacode.h: balanced Gold codes of period 1023, PRN 1 starts 1100100000 (IS-GPS-200 octal 1440)."""
    L = lib("f64")
    chips = np.zeros((32, 1023), np.uint8)
    for sv in range(32):
        L.oracle_ca_chips(sv, _p(chips[sv]))
    assert list(chips[0][:10]) == [1, 1, 0, 0, 1, 0, 0, 0, 0, 0]
    assert all(int(c.sum()) == 512 for c in chips)
    b = 1.0 - 2.0 * chips
    # Gold-code cross-correlation takes only the three values -65, -1, 63
    cc = np.fft.ifft(np.fft.fft(b[0]) * np.conj(np.fft.fft(b[7]))).real.round().astype(int)
    assert set(cc) <= {-65, -1, 63}
    ac = np.fft.ifft(np.abs(np.fft.fft(b[20])) ** 2).real.round().astype(int)
    assert ac[0] == 1023 and set(ac[1:]) <= {-65, -1, 63}
    # SearchCode(): G1 is all ones after 0 chips and again after 1023
    assert L.oracle_search_code(3, 0x3FF) == 0


def _table_3_I(golden_dir):
    t = json.load(open(os.path.join(golden_dir, "ref_known_answers.json")))["is_gps_200_table_3_I"]
    first10 = [[(int(o, 8) >> (9 - i)) & 1 for i in range(10)] for o in t["first_10_chips_octal"]]
    return first10, t["g2_delay_chips"]


def _g1_g2():
    """G1 = 1 + x^3 + x^10, G2 = 1 + x^2 + x^3 + x^6 + x^8 + x^9 + x^10, all-ones start (IS-GPS-200 3.3.2.3): the two
    maximal-length sequences by themselves, written here independently of the oracle and of the product."""
    g1, g2, o1, o2 = [1] * 10, [1] * 10, [], []
    for _ in range(1023):
        o1.append(g1[9])
        o2.append(g2[9])
        g1 = [g1[2] ^ g1[9]] + g1[:9]
        g2 = [g2[1] ^ g2[2] ^ g2[5] ^ g2[7] ^ g2[8] ^ g2[9]] + g2[:9]
    return np.array(o1, np.uint8), np.array(o2, np.uint8)


def test_all_32_prns_against_is_gps_200_table_3_I(golden_dir):
    """The reference-held definition of the codes (IS-GPS-200G.pdf, Table 3-I, in the reference repository): the first ten chips
    of every PRN in octal, and the G2 delay of every PRN -- all 1023 chips = G1 xor G2 delayed by that many chips.  Pins the tap
    table of c/search_offline.cpp:20-53 and the generator of c/cacode.h:9-35 as the oracle restates them, for every PRN."""
    first10, delays = _table_3_I(golden_dir)
    assert len(first10) == 32 and len(delays) == 32
    L = lib("f64")
    g1, g2 = _g1_g2()
    for sv in range(32):
        chips = np.zeros(1023, np.uint8)
        L.oracle_ca_chips(sv, _p(chips))
        assert list(chips[:10]) == first10[sv], f"PRN {sv + 1}"
        assert np.array_equal(chips, g1 ^ np.roll(g2, delays[sv])), f"PRN {sv + 1}"


def test_grid_sizes():
    L = lib("f64")
    L.oracle_dmax.restype = int
    assert L.oracle_dmax(5.456e6, 5000.0) == 36 and L.oracle_nlags(5.456e6) == 5456
    assert L.oracle_dmax(8.184e6, 5000.0) == 24 and L.oracle_nlags(8.184e6) == 8184
    assert L.oracle_dmax(2.8e6, 5000.0) == 71 and L.oracle_nlags(2.8e6) == 2800
    assert L.oracle_dmax(2.8e6, 100000.0) == 1428
    assert L.oracle_nlags(2.8001e6) == 2801  # i < FS/1000 with a fractional bound


@pytest.mark.parametrize("name,fc,fs,file", [("nott", 4.092e6, 5.456e6, "synth_nott_fs5456.bin"),
                                             ("sigtmp", 2.046e6, 8.184e6, "gps_sig_tmp.bin"),
                                             ("rtl", 0.62e6, 2.8e6, "synth_rtl_fs2800.bin")])
def test_cells_vs_numpy_golden(golden_dir, name, fc, fs, file):
    z = np.load(os.path.join(golden_dir, f"np64_cells_{name}.npz"))
    buf = open(os.path.join(golden_dir, file), "rb").read()
    orc = Oracle(fc, fs, float(z["max_fo"]), ref_quirks=bool(z["quirks"]))
    assert orc.dmax == int(z["dmax"]) and orc.num_lags == int(z["S"])
    pairs = [tuple(int(v) for v in p) for p in z["pairs"]][:3]
    for b, sv in pairs:
        cells, _ = orc.search_block(buf[b * 5120:(b + 1) * 5120], sv)
        np.testing.assert_allclose(cells["max_pwr"], z[f"max_pwr_{b}_{sv}"], rtol=5e-6)
        np.testing.assert_allclose(cells["tot_pwr"], z[f"tot_pwr_{b}_{sv}"], rtol=5e-6)
        assert np.array_equal(cells["max_i"], z[f"max_i_{b}_{sv}"])


def test_survey_transcript_gps_sig_tmp(golden_dir):
    """gps_test gps_sig_tmp.bin 2.046e6 8.184e6 5000: the oracle against the survey's transcript -- the
    survey's MKL-shim build of the reference sources, a STAND-IN (no FFTW in the image): it pins nothing by itself and is kept
    as a regression transcript; the pins of this file are test_code_phase_follows_from_gps_sig_gen (the generator script's own
    parameters) and the README's known answer below.  First 3 runs here (the full 12 run in the GPU suite)."""
    known = json.load(open(os.path.join(golden_dir, "ref_known_answers.json")))["gps_sig_tmp"]
    orc = Oracle(2.046e6, 8.184e6, 5000.0, ref_quirks=True)
    n, text, peaks = orc.search_file(os.path.join(golden_dir, "gps_sig_tmp.bin"), max_runs=3)
    assert n == 3
    lines = text.split("\n")
    assert lines[0].split() == ["0", "satellite:"] + [str(v) for v in known["run0_hits_sv"]]
    assert lines[1].split()[2:] == ["%.1f" % v for v in known["run0_hits_snr"]]
    assert lines[2].split()[2:] == [str(v) for v in known["run0_hits_lo"]]
    assert lines[3].split()[2:] == [str(v) for v in known["run0_hits_ca"]]
    sv7 = peaks[7::32]
    assert ["%.1f" % v for v in sv7["snr"]] == ["%.1f" % v for v in known["sv7_snr"][:3]]
    assert list(sv7["lo_shift"]) == known["sv7_lo_shift"][:3]
    assert list(sv7["ca_shift"]) == known["sv7_ca_shift"][:3]
    others = np.delete(peaks["snr"], [7, 39, 71])
    assert others.min() > known["other_best_snr_range"][0] - 0.5 and others.max() < known["other_best_snr_range"][1] + 0.5
    # README.md:45,57 known answer: PRN 8 at zero Doppler is the signal in the file
    assert int(np.argmax(peaks["snr"][:32])) == 7 and int(peaks["lo_shift"][7]) == 0


def test_code_phase_follows_from_gps_sig_gen(golden_dir):
    """A known answer that needs no reference run: README.md:57 ("C/A codes results are aligned with GPS signal we
    generate") made exact.  gps_sig_gen.m starts chip 0 of PRN 8 at sample 0 of its impulse train and shapes it with
    rcosine(1, 8) (delay 3 chips = 24 samples, :22,35): chip k peaks at sample 24 + 8 k, i.e. its leading edge -- what the
    search's zero-order-hold replica aligns to -- is at 20 + 8 k.  SearchTask gives PRN index 7 the blocks 32 r + 7, which
    start at sample 40960 (32 r + 7) (10 x 512 bytes per Sample() call, :129,135-136), so with 8184 samples per code
    period Correlate must report ca_shift = (40960 (32 r + 7) - 20) mod 8184 and Doppler bin 0 (+-1: the 204.6 Hz bins
    straddle zero).  That is 260, 1540, 2820, ... -- the sequence the survey's reference run printed."""
    expect = [(40960 * (32 * r + 7) - 20) % 8184 for r in range(12)]
    assert expect[:4] == [260, 1540, 2820, 4100]
    buf = open(os.path.join(golden_dir, "gps_sig_tmp.bin"), "rb").read()
    orc = Oracle(2.046e6, 8.184e6, 5000.0)
    for r in (0, 5, 11):
        b = 32 * r + 7
        _, pk = orc.search_block(buf[b * 5120:(b + 1) * 5120], 7)
        assert int(pk["ca_shift"]) == expect[r] and abs(int(pk["lo_shift"])) <= 1 and pk["snr"] > 500


def test_every_block_of_gps_sig_tmp_has_prn8_where_gps_sig_gen_puts_it(golden_dir):
    """The same derivation at EVERY block b of the file, searched for PRN 8 (384 integer pins instead of 12):
    ca_shift = (40960 b - 20) mod 8184, Doppler bin 0 +- 1.  (The GPU counterpart is in tests/test_gpu_parity.py.)"""
    buf = open(os.path.join(golden_dir, "gps_sig_tmp.bin"), "rb").read()
    orc = Oracle(2.046e6, 8.184e6, 5000.0, kind="f32")
    for b in range(384):
        _, pk = orc.search_block(buf[b * 5120:(b + 1) * 5120], 7)
        assert int(pk["ca_shift"]) == (40960 * b - 20) % 8184, b
        assert abs(int(pk["lo_shift"])) <= 1 and pk["snr"] > 300, (b, pk)


def test_quirk_only_touches_prn_index_0(golden_dir):
    buf = open(os.path.join(golden_dir, "gps_sig_tmp.bin"), "rb").read()[:32 * 5120]
    a = Oracle(2.046e6, 8.184e6, 5000.0, ref_quirks=True)
    b = Oracle(2.046e6, 8.184e6, 5000.0, ref_quirks=False)
    for blk, sv in ((0, 0), (5, 5)):
        ca, pa = a.search_block(buf[blk * 5120:(blk + 1) * 5120], sv)
        cb, pb = b.search_block(buf[blk * 5120:(blk + 1) * 5120], sv)
        same = np.array_equal(ca, cb)
        assert same == (sv != 0)


def test_short_file_and_missing_file(tmp_path, golden_dir):
    orc = Oracle(4.092e6, 5.456e6, 5000.0)
    n, text, _ = orc.search_file(str(tmp_path / "nope.bin"))
    assert n == -1 and text == "can not open file!\n"
    p = tmp_path / "short.bin"
    p.write_bytes(open(os.path.join(golden_dir, "synth_nott_fs5456.bin"), "rb").read()[:31 * 5120 + 100])
    n, text, _ = orc.search_file(str(p))
    assert n == 0 and text == "run out of file!\n"


def test_f32_port_close_to_f64(golden_dir):
    buf = open(os.path.join(golden_dir, "synth_nott_fs5456.bin"), "rb").read()
    a = Oracle(4.092e6, 5.456e6, 5000.0, kind="f64")
    b = Oracle(4.092e6, 5.456e6, 5000.0, kind="f32")
    ca, pa = a.search_block(buf[20 * 5120:21 * 5120], 20)
    cb, pb = b.search_block(buf[20 * 5120:21 * 5120], 20)
    np.testing.assert_allclose(ca["max_pwr"], cb["max_pwr"], rtol=2e-4)
    assert pa["ca_shift"] == pb["ca_shift"] and pa["lo_shift"] == pb["lo_shift"]


@pytest.mark.parametrize("real", ["double", "float"])
def test_oracle_is_clean_under_asan_and_ubsan(golden_dir, tmp_path, real):
    """the reference overruns fwd_buf (c/search_offline.cpp:135-157); the restatement emulates that
    explicitly and is itself free of it.  tests/c/oracle_san.c walks every entry point of oracle/gpsacq_oracle.c (both quirk
    modes, a short and a missing file, a report buffer that is too small) under -fsanitize=address,undefined."""
    import shutil
    import subprocess
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    exe = str(tmp_path / "oracle_san")
    cmd = ["gcc", "-O1", "-g", "-std=c11", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
           "-DORACLE_REAL=" + real] + (["-fopenmp"] if real == "float" else []) + ["-o", exe, os.path.join(ROOT, "tests", "c", "oracle_san.c"), "-lm"]
    b = subprocess.run(cmd, capture_output=True, text=True)
    if b.returncode != 0 and "sanitize" in b.stderr + b.stdout and ("cannot find" in b.stderr or "not supported" in b.stderr):
        pytest.skip("sanitizer runtime not installed")
    assert b.returncode == 0, b.stderr
    r = subprocess.run([exe, os.path.join(golden_dir, "gps_sig_tmp.bin"), "2.046e6", "8.184e6"], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="print_stacktrace=1"))
    assert r.returncode == 0 and "oracle_san: ok" in r.stdout, r.stdout + r.stderr


def test_oracle_cells_against_an_independent_float32_fft(golden_dir):
    """How far can float rounding of the transforms move a cell?  The reference runs FFTW in single precision (absent here); the
    checker transforms in double.  The same cells with every transform done by a third implementation in float32 (scipy's
    pocketfft on complex64) differ from the checker's by < 3e-6 relative in max_pwr and tot_pwr with the same argmax -- thirty
    times inside north_star's 1e-4 -- and the SNR that decides the threshold-25 hit list moves by < 3e-6 too."""
    import scipy.fft
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from make_golden import code_replica
    N = 40000
    for name, fc, fs, picks in (("gps_sig_tmp.bin", 2.046e6, 8.184e6, [(7, 7), (0, 0), (40, 8)]), ("synth_nott_fs5456.bin", 4.092e6, 5.456e6, [(0, 0), (20, 20)])):
        buf = np.fromfile(os.path.join(golden_dir, name), dtype=np.uint8)
        orc = Oracle(fc, fs, 5000.0)
        quad = np.zeros(40960, np.uint8)
        lib().oracle_lo_quadrants(fc, fs, 40960, _p(quad))
        for b, sv in picks:
            blk = np.ascontiguousarray(buf[b * 5120:(b + 1) * 5120])
            cells, _ = orc.search_block(blk, sv)
            mixed = np.zeros(2 * 40960, np.float32)
            lib().oracle_mix_block(_p(blk), _p(quad), _p(mixed))
            D = scipy.fft.fft(mixed.view(np.complex64)[:N])
            C = scipy.fft.fft(np.asarray(code_replica(fs, sv), dtype=np.float32).astype(np.complex64))
            assert D.dtype == np.complex64 and C.dtype == np.complex64
            for d in range(-orc.dmax, orc.dmax + 1):
                y = scipy.fft.ifft((np.conj(D) * np.roll(C, d)).astype(np.complex64)) * np.float32(N)
                assert y.dtype == np.complex64
                pwr = (y.real * y.real + y.imag * y.imag)[:orc.num_lags]
                c = cells[d + orc.dmax]
                tot = np.cumsum(pwr, dtype=np.float32)[-1]  # the reference's own float accumulation, lag by lag (:193)
                assert abs(pwr.max() / c["max_pwr"] - 1) < 3e-6 and abs(float(tot) / c["tot_pwr"] - 1) < 3e-6
                snr = pwr.max() / (tot / np.float32(orc.num_lags))
                assert abs(snr / c["snr"] - 1) < 3e-6
                assert int(pwr.argmax()) == c["max_i"] or abs(pwr[c["max_i"]] / pwr.max() - 1) < 1e-5
