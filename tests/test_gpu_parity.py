"""GPU parity: the HIP path (through the C ABI) against the CPU oracle and the golden fixtures.

Tolerances (BASELINE.json north_star): code-phase bin identical, Doppler within one bin,
correlator magnitudes within 1e-4 relative.  We test tighter where the arithmetic allows:
per-cell max_pwr / tot_pwr to 2e-5 relative, identical argmax unless the two top powers of a
cell are closer than 1e-5 relative (a float-rounding tie).
"""
import json
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

REL = 2e-5


def _blocks(buf):
    return [bytes(buf[i * 5120:(i + 1) * 5120]) for i in range(len(buf) // 5120)]


@pytest.fixture(scope="module")
def gpsacq_mod():
    import gpsacq
    gpsacq.load_library()
    return gpsacq


CONFIGS = {
    "nott": dict(fc=4.092e6, fs=5.456e6, file="synth_nott_fs5456.bin"),
    "sigtmp": dict(fc=2.046e6, fs=8.184e6, file="gps_sig_tmp.bin"),
    "rtl": dict(fc=0.62e6, fs=2.8e6, file="synth_rtl_fs2800.bin"),
}


def _cmp_cells(gc, oc, what, tie=None):
    """gc, oc: structured arrays [tasks, bins].  Magnitudes to REL; the argmax (code-phase bin) must be IDENTICAL unless
    the oracle's own powers at the two lags differ by less than 1e-5 relative -- a float-rounding tie, which `tie(t, d)`
    (returning the oracle's per-lag powers of cell (t, d)) is asked to prove for every mismatch."""
    np.testing.assert_allclose(gc["max_pwr"], oc["max_pwr"], rtol=REL, err_msg=what + " max_pwr")
    np.testing.assert_allclose(gc["tot_pwr"], oc["tot_pwr"], rtol=REL, err_msg=what + " tot_pwr")
    np.testing.assert_allclose(gc["snr"], oc["snr"], rtol=2 * REL, err_msg=what + " snr")
    bad = np.argwhere(gc["max_i"] != oc["max_i"])
    assert len(bad) <= 3, f"{what}: {len(bad)} argmax mismatches of {gc.size}"
    for t, d in bad:
        assert tie is not None, f"{what}: argmax mismatch at task {t} bin {d} and no tie check available"
        pw = tie(int(t), int(d))
        a, b = float(pw[gc["max_i"][t, d]]), float(pw[oc["max_i"][t, d]])
        assert abs(a - b) <= 1e-5 * b, f"{what}: task {t} bin {d}: lags {gc['max_i'][t, d]} / {oc['max_i'][t, d]} are not a tie ({a} vs {b})"


def _tie_fn(orc, buf, tasks):
    """per-lag powers of the oracle for cell (task index, bin index)"""
    def f(t, d):
        b, sv = tasks[t]
        orc.L.oracle_sample(orc.h, _np_ptr(np.frombuffer(buf[b * 5120:(b + 1) * 5120], dtype=np.uint8).copy()))
        pw = np.zeros(orc.num_lags, np.float32)
        orc.L.oracle_cell_power(orc.h, sv, d - orc.dmax, _np_ptr(pw))
        return pw
    return f


def _np_ptr(a):
    import ctypes
    f = _np_ptr
    f.keep = a  # the array must outlive the call
    return a.ctypes.data_as(ctypes.c_void_p)


@pytest.mark.parametrize("name", ["nott", "sigtmp", "rtl"])
def test_code_and_sample_spectra(gpsacq_mod, golden_dir, name):
    from oracle_lib import Oracle
    cfg = CONFIGS[name]
    buf = open(os.path.join(golden_dir, cfg["file"]), "rb").read()
    with gpsacq_mod.Engine(cfg["fc"], cfg["fs"], 5000.0) as eng:
        orc = Oracle(cfg["fc"], cfg["fs"], 5000.0)
        assert eng.dmax == orc.dmax and eng.num_lags == orc.num_lags
        for sv in (0, 7, 20, 31):
            g, o = eng.code_spectrum(sv), orc.code_spectrum(sv)
            scale = np.abs(o).max()
            assert np.abs(g - o).max() / scale < 2e-6, f"code spectrum sv {sv}"
        for b in (0, 5):
            blk = buf[b * 5120:(b + 1) * 5120]
            g, o = eng.sample_spectrum(blk), orc.sample_spectrum(blk)
            scale = np.abs(o).max()
            assert np.abs(g - o).max() / scale < 2e-6, f"sample spectrum block {b}"


@pytest.mark.parametrize("name", ["nott", "sigtmp", "rtl"])
def test_cells_vs_torch_fp32_reference(gpsacq_mod, golden_dir, name):
    """The whole cell computation against a plain float32 restatement on the device whose transforms are the vendor library's
    (torch.fft = rocFFT, complex64): Sample()'s XOR-mixed samples (from the oracle's mixer, exact +-1 values) -> fft ->
    conj(data) * shifted code spectrum -> ifft * N -> |.|^2 over the first S lags -> max / argmax / sum (c/search_offline.cpp:161,
    176-196).  The reference's own transforms are FFTW in float, which this image does not have; a second, independent
    single-precision FFT is the nearest thing available here, and north_star's bar for the FFTW path (magnitudes within 1e-4
    relative) is met against it with a factor of ten to spare."""
    import torch
    from oracle_lib import lib, _p
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from make_golden import code_replica
    cfg = CONFIGS[name]
    fc, fs = cfg["fc"], cfg["fs"]
    buf = np.frombuffer(open(os.path.join(golden_dir, cfg["file"]), "rb").read(), dtype=np.uint8)
    quad = np.zeros(40960, np.uint8)
    lib().oracle_lo_quadrants(fc, fs, 40960, _p(quad))
    tasks = [(0, 0), (1, 20), (2, 7), (3, 31), (4, 28)]
    N = 40000
    with gpsacq_mod.Engine(fc, fs, 5000.0) as eng:
        cells, _ = eng.search(buf[:5 * 5120], tasks=tasks)
        S, dmax = eng.num_lags, eng.dmax
    worst = 0.0
    for t, (b, sv) in enumerate(tasks):
        blk = np.ascontiguousarray(buf[b * 5120:(b + 1) * 5120])
        mixed = np.zeros(2 * 40960, np.float32)
        lib().oracle_mix_block(_p(blk), _p(quad), _p(mixed))
        x = torch.from_numpy(mixed.view(np.complex64)[:N].copy()).cuda()
        D = torch.fft.fft(x)
        C = torch.fft.fft(torch.from_numpy(np.asarray(code_replica(fs, sv), dtype=np.float32)).cuda().to(torch.complex64))
        assert D.dtype == torch.complex64 and C.dtype == torch.complex64
        for d in range(-dmax, dmax + 1):
            y = torch.fft.ifft(torch.conj(D) * torch.roll(C, d)) * N  # rev_buf[i] = conj(data[i]) * code[(i - dop) mod N]; backward, unnormalised
            pwr = (y.real * y.real + y.imag * y.imag)[:S]
            got = cells[t][d + dmax]
            mx, mi = float(pwr.max()), int(pwr.argmax())
            tot = float(pwr.sum(dtype=torch.float64))
            worst = max(worst, abs(got["max_pwr"] / mx - 1.0), abs(got["tot_pwr"] / tot - 1.0))
            if got["max_i"] != mi:  # two lags within float rounding of each other
                assert abs(float(pwr[got["max_i"]]) / mx - 1.0) < 1e-5, (t, d, got["max_i"], mi)
    print(f"worst relative difference to the rocFFT float32 restatement ({name}): {worst:.2e}")
    assert worst < 1e-5, worst


@pytest.mark.parametrize("name", ["nott", "sigtmp", "rtl"])
def test_cells_vs_oracle(gpsacq_mod, golden_dir, name):
    from oracle_lib import Oracle
    cfg = CONFIGS[name]
    buf = open(os.path.join(golden_dir, cfg["file"]), "rb").read()
    nblk = 33
    buf = buf[:nblk * 5120]
    with gpsacq_mod.Engine(cfg["fc"], cfg["fs"], 5000.0) as eng:
        gcells, gpeaks = eng.search(buf)
        orc = Oracle(cfg["fc"], cfg["fs"], 5000.0)
        tasks = [(b, b % 32) for b in range(nblk)]  # all 33 tasks: one whole run plus the first block of the next
        ocells, opeaks = orc.search(buf, tasks)
        _cmp_cells(gcells, ocells, name, _tie_fn(orc, buf, tasks))
        assert np.array_equal(gpeaks["ca_shift"], opeaks["ca_shift"])
        assert np.array_equal(gpeaks["lo_shift"], opeaks["lo_shift"])
        np.testing.assert_allclose(gpeaks["snr"], opeaks["snr"], rtol=1e-4)


def test_full_run_with_reference_quirk(gpsacq_mod, golden_dir):
    """One complete run of the reference schedule (32 PRN x 73 bins) at fs 5.456 MHz with ref_quirks = 1: PRN index 0 is
    correlated against code[0] with its first 960 bins replaced by the block's samples 40000..40959, as a real gps_test
    binary does; every cell against the oracle run the same way."""
    from oracle_lib import Oracle
    buf = open(os.path.join(golden_dir, "synth_nott_fs5456.bin"), "rb").read()[:32 * 5120]
    tasks = [(b, b) for b in range(32)]
    with gpsacq_mod.Engine(4.092e6, 5.456e6, 5000.0, ref_quirks=True) as eng:
        gcells, gpeaks = eng.search(buf)
    orc = Oracle(4.092e6, 5.456e6, 5000.0, ref_quirks=True)
    ocells, opeaks = orc.search(buf, tasks)
    # tie helper on the SAME (quirk) oracle: oracle_sample() of the task's block re-applies that block's patch to code[0][0..959]
    # (c/search_offline.cpp:61,135-157) before oracle_cell_power() reads it, so PRN index 0's per-lag powers are the patched ones
    _cmp_cells(gcells, ocells, "quirk run", _tie_fn(orc, buf, tasks))
    assert np.array_equal(gpeaks["ca_shift"], opeaks["ca_shift"]) and np.array_equal(gpeaks["lo_shift"], opeaks["lo_shift"])
    # the quirk matters: without it PRN index 0 reads differently
    with gpsacq_mod.Engine(4.092e6, 5.456e6, 5000.0) as eng:
        plain, _ = eng.search(buf[:5120])
    assert not np.allclose(plain["max_pwr"][0], gcells["max_pwr"][0], rtol=1e-3)


@pytest.mark.parametrize("name,npz", [("nott", "np64_cells_nott.npz"), ("sigtmp", "np64_cells_sigtmp.npz"), ("rtl", "np64_cells_rtl.npz")])
def test_cells_vs_numpy_golden(gpsacq_mod, golden_dir, name, npz):
    """Committed vectors of the independent float64 numpy restatement (make_golden.py)."""
    cfg = CONFIGS[name]
    z = np.load(os.path.join(golden_dir, npz))
    buf = open(os.path.join(golden_dir, cfg["file"]), "rb").read()
    pairs = [tuple(p) for p in z["pairs"]]
    with gpsacq_mod.Engine(cfg["fc"], cfg["fs"], float(z["max_fo"]), ref_quirks=bool(z["quirks"])) as eng:
        assert eng.dmax == int(z["dmax"]) and eng.num_lags == int(z["S"])
        cells, _ = eng.search(buf, tasks=pairs)
        for t, (b, sv) in enumerate(pairs):
            np.testing.assert_allclose(cells["max_pwr"][t], z[f"max_pwr_{b}_{sv}"], rtol=REL)
            np.testing.assert_allclose(cells["tot_pwr"][t], z[f"tot_pwr_{b}_{sv}"], rtol=REL)
            bad = np.flatnonzero(cells["max_i"][t] != z[f"max_i_{b}_{sv}"])
            assert len(bad) <= 1
            for d in bad:  # only a rounding tie may differ: the GPU's lag must be the golden runner-up and the two golden powers a tie
                assert cells["max_i"][t][d] == z[f"second_i_{b}_{sv}"][d], f"task {t} bin {d}: lag {cells['max_i'][t][d]} is not the runner-up"
                assert abs(z[f"second_pwr_{b}_{sv}"][d] / z[f"max_pwr_{b}_{sv}"][d] - 1) < 1e-5, f"task {t} bin {d}: not a tie"


def _edge_tokens(snrs):
    """printf renderings an SNR may legitimately take when it sits on a rounding edge of the report's two formats
    (%5.1f in the hit list, %2.0f in the all-PRN line, c/search_offline.cpp:270-287): both neighbours of the edge."""
    out = set()
    for s in np.asarray(snrs, dtype=np.float64):
        if abs(s * 10 - np.floor(s * 10) - 0.5) < 0.02:
            out |= {"%.1f" % (np.floor(s * 10) / 10), "%.1f" % (np.floor(s * 10) / 10 + 0.1)}
        if abs(s - np.floor(s) - 0.5) < 0.002:
            out |= {"%.0f" % np.floor(s), "%.0f" % (np.floor(s) + 1)}
    return out


def assert_report_equal_up_to_printf_edges(got, want, opeaks):
    """SearchTask report `got` against the oracle's `want`: identical, or at most 3 lines differ, each in ONE token, and
    that token is the rendering of an oracle SNR of the SAME run that sits on a printf rounding edge (both texts must
    then show the two neighbouring renderings of that value)."""
    if got == want:
        return
    a, b = got.split("\n"), want.split("\n")
    assert len(a) == len(b), "reports differ in their number of lines"
    diff = [(i, x, y) for i, (x, y) in enumerate(zip(a, b)) if x != y]
    assert len(diff) <= 3, diff
    for i, x, y in diff:
        xs, ys = x.split(), y.split()
        assert len(xs) == len(ys), (x, y)
        tok = [(p, q) for p, q in zip(xs, ys) if p != q]
        assert len(tok) == 1, (x, y)
        run = i // 6  # six lines per run (:264-287)
        edges = _edge_tokens(opeaks["snr"][run * 32:(run + 1) * 32])
        p_, q_ = tok[0]
        assert p_ in edges and q_ in edges, f"line {i}: {p_!r} vs {q_!r} is not a printf edge of run {run} ({sorted(edges)})"


def test_gps_sig_tmp_known_answers(gpsacq_mod, golden_dir):
    """gps_test gps_sig_tmp.bin 2.046e6 8.184e6 5000 with ref_quirks.  What is PINNED here comes from the reference's own generator
    script (PRN 8, Doppler bin 0, ca_shift = (40960 b - 20) mod 8184: gps_sig_gen.m's parameters) and from the oracle, line by
    line a restatement of c/search_offline.cpp.  tests/golden/ref_known_answers.json is the survey's MKL-shim transcript -- a
    stand-in build of the reference sources, which pins nothing by itself; it is compared as a regression transcript only."""
    from oracle_lib import Oracle
    known = json.load(open(os.path.join(golden_dir, "ref_known_answers.json")))["gps_sig_tmp"]
    path = os.path.join(golden_dir, "gps_sig_tmp.bin")
    buf = open(path, "rb").read()
    nrun = len(buf) // 5120 // 32
    assert nrun == known["runs"]
    buf = buf[:nrun * 32 * 5120]
    with gpsacq_mod.Engine(2.046e6, 8.184e6, 5000.0, ref_quirks=True) as eng:
        _, peaks = eng.search(buf, want_cells=False)
    sv7 = peaks[7::32]
    # derived from gps_sig_gen.m alone (tests/test_oracle.py::test_code_phase_follows_from_gps_sig_gen): all 12 runs
    assert list(sv7["ca_shift"]) == [(40960 * (32 * r + 7) - 20) % 8184 for r in range(12)]
    assert list(sv7["lo_shift"]) == known["sv7_lo_shift"]
    assert list(sv7["ca_shift"]) == known["sv7_ca_shift"]
    assert ["%.1f" % s for s in sv7["snr"]] == ["%.1f" % s for s in known["sv7_snr"]]
    run0 = peaks[:32]
    hits = [sv for sv in range(32) if not run0["snr"][sv] < 25]
    assert hits == known["run0_hits_sv"]
    assert [int(run0["lo_shift"][sv]) for sv in hits] == known["run0_hits_lo"]
    assert [int(run0["ca_shift"][sv]) for sv in hits] == known["run0_hits_ca"]
    assert ["%.1f" % run0["snr"][sv] for sv in hits] == ["%.1f" % s for s in known["run0_hits_snr"]]
    # whole report against the oracle's SearchTask restatement (guard band: SNRs within 0.02 of a
    # printf rounding edge or of the threshold may legitimately differ in the last digit)
    orc = Oracle(2.046e6, 8.184e6, 5000.0, ref_quirks=True)
    n, text, opeaks = orc.search_file(path)
    assert n == nrun
    assert np.array_equal(peaks["ca_shift"], opeaks["ca_shift"])
    assert np.array_equal(peaks["lo_shift"], opeaks["lo_shift"])
    np.testing.assert_allclose(peaks["snr"], opeaks["snr"], rtol=1e-4)
    report = gpsacq_mod.format_report(peaks) + "run out of file!\n"
    assert_report_equal_up_to_printf_edges(report, text, opeaks)


def test_every_block_of_gps_sig_tmp_finds_prn8_where_gps_sig_gen_puts_it(gpsacq_mod, golden_dir):
    """384 integer pins from the reference's own generator instead of 12: gps_sig_gen.m (PRN 8, no Doppler, 8 samples per
    chip, raised-cosine delay 24 samples minus half a chip) fixes the code phase at EVERY 40960-sample block b of
    gps_sig_tmp.bin, not only at the blocks 32 r + 7 the reference schedule pairs with PRN 8:
    ca_shift = (40960 b - 20) mod 8184, Doppler bin 0 +- 1 (a navigation-bit flip inside a block splits the peak)."""
    buf = open(os.path.join(golden_dir, "gps_sig_tmp.bin"), "rb").read()
    nblk = len(buf) // 5120 // 32 * 32
    assert nblk == 384
    with gpsacq_mod.Engine(2.046e6, 8.184e6, 5000.0) as eng:
        _, peaks = eng.search(buf[:nblk * 5120], tasks=[(b, 7) for b in range(nblk)], want_cells=False)
    assert list(peaks["ca_shift"]) == [(40960 * b - 20) % 8184 for b in range(nblk)]
    assert np.abs(peaks["lo_shift"]).max() <= 1
    assert peaks["snr"].min() > 300


def test_nottingham_standin_known_prns(gpsacq_mod, golden_dir):
    """Synthetic stand-in for the absent Nottingham capture: the five PRNs of the JKS table
    are injected with the table's Doppler bins and code phases."""
    known = json.load(open(os.path.join(golden_dir, "ref_known_answers.json")))["nottingham_jks_table"]
    buf = open(os.path.join(golden_dir, "synth_nott_fs5456.bin"), "rb").read()
    with gpsacq_mod.Engine(4.092e6, 5.456e6, 5000.0) as eng:
        _, peaks = eng.search(buf[:32 * 5120], want_cells=False)
    hits = [sv + 1 for sv in range(32) if peaks["snr"][sv] >= 25]
    assert hits == known["prn"]
    for prn, lo, ca in zip(known["prn"], known["lo_shift"], known["ca_shift"]):
        sv = prn - 1
        assert abs(int(peaks["lo_shift"][sv]) - lo) <= 1
        fd = lo * 5.456e6 / 40000
        expect = (ca + 40960 * sv * (1 + fd / 1575.42e6)) % 5456
        got = int(peaks["ca_shift"][sv])
        assert min(abs(got - expect), 5456 - abs(got - expect)) <= 1.5


def test_all_prn_grid_and_linearity(gpsacq_mod, golden_dir):
    """Acquisition grid (one block against all 32 PRNs) equals 32 single-PRN searches, and a
    batch equals its halves (size-independent properties at batch sizes the oracle cannot reach)."""
    buf = open(os.path.join(golden_dir, "synth_nott_fs5456.bin"), "rb").read()
    with gpsacq_mod.Engine(4.092e6, 5.456e6, 5000.0) as eng:
        tasks = [(0, sv) for sv in range(32)] + [(3, sv) for sv in range(32)]
        cells, peaks = eng.search(buf, tasks=tasks)
        c2, p2 = eng.search(buf, tasks=tasks[32:])
        assert np.array_equal(cells[32:], c2) and np.array_equal(peaks[32:], p2)
        full_c, full_p = eng.search(buf)
        a_c, a_p = eng.search(buf[:20 * 5120])
        assert np.array_equal(full_c[:20], a_c) and np.array_equal(full_p[:20], a_p)
        # bit-reproducible across calls
        again_c, again_p = eng.search(buf)
        assert np.array_equal(full_c, again_c) and np.array_equal(full_p, again_p)


def test_large_doppler_range(gpsacq_mod, golden_dir):
    """max_fo honoured (the reference CLI ignores argv[4], its library does not): +-100 kHz at
    fs 2.8 MHz = 2857 bins; the +-5 kHz sub-range must reproduce the +-5 kHz search."""
    buf = open(os.path.join(golden_dir, "synth_rtl_fs2800.bin"), "rb").read()[:8 * 5120]
    with gpsacq_mod.Engine(0.62e6, 2.8e6, 5000.0) as e1, gpsacq_mod.Engine(0.62e6, 2.8e6, 100000.0) as e2:
        c1, _ = e1.search(buf)
        c2, p2 = e2.search(buf)
        assert e2.num_doppler == 2 * 1428 + 1
        lo = e2.dmax - e1.dmax
        assert np.array_equal(c1, c2[:, lo:lo + e1.num_doppler])
        assert int(p2["lo_shift"][7]) == 20


def test_errors(gpsacq_mod):
    with pytest.raises(gpsacq_mod.GpsAcqError):
        gpsacq_mod.Engine(1e6, 5e6, 3e6)  # Doppler range beyond half the sampling rate
    with gpsacq_mod.Engine(4.092e6, 5.456e6, 5000.0) as eng:
        with pytest.raises(ValueError):
            eng.search(b"\x00" * 100)
        with pytest.raises(gpsacq_mod.GpsAcqError):
            eng.search(b"\x00" * 5120, tasks=[(3, 0)])
        with pytest.raises(gpsacq_mod.GpsAcqError):
            eng.search(b"\x00" * 5120, tasks=[(0, 32)])
        # all-zero bits are a valid capture (constant +1 samples): must not produce NaN
        cells, peaks = eng.search(b"\x00" * 5120)
        assert np.isfinite(cells["snr"]).all() and np.isfinite(peaks["snr"]).all()


def test_gps_test_cli_stdout(golden_dir):
    """The C++ front end (same argv handling and printf formats as c/test_search_offline.cpp +
    SearchTask) against the oracle's SearchTask text on the bundled capture, reference quirk on."""
    import subprocess
    from oracle_lib import Oracle
    from test_host import BANNER, GPS_TEST
    path = os.path.join(golden_dir, "gps_sig_tmp.bin")
    env = dict(os.environ, GPSACQ_REF_QUIRKS="1", GPSACQ_BATCH_RUNS="5")
    r = subprocess.run([GPS_TEST, path, "2.046e6", "8.184e6", "5000"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr
    assert r.stdout.startswith(BANNER)
    body = r.stdout[len(BANNER):]
    orc = Oracle(2.046e6, 8.184e6, 5000.0, ref_quirks=True)
    n, text, opeaks = orc.search_file(path)
    assert body.endswith("run out of file!\n") and body.count("satellite:") == n == 12
    assert_report_equal_up_to_printf_edges(body, text, opeaks)
    # missing file: same message as the reference, exit code 0
    r = subprocess.run([GPS_TEST, "/nonexistent.bin", "2.046e6", "8.184e6", "5000"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout == BANNER + "can not open file!\n"


def test_reference_main_against_our_library(golden_dir):
    """The reference's UNMODIFIED front end (c/test_search_offline.cpp, compiled in the authoring container by `make
    dropin-check` into oracle/_ref/gps_test_refmain) linked against libgps_search.so / libgpsacq.so: same stdout as our own
    gps_test -- the drop-in boundary of SURVEY.md section 8(b) exercised end to end."""
    import subprocess
    from test_host import GPS_TEST, ROOT
    exe = os.path.join(ROOT, "oracle", "_ref", "gps_test_refmain")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/gps_test_refmain not built (make dropin-check needs /root/reference)")
    path = os.path.join(golden_dir, "gps_sig_tmp.bin")
    env = dict(os.environ, GPSACQ_REF_QUIRKS="1")
    a = subprocess.run([exe, path, "2.046e6", "8.184e6", "5000"], capture_output=True, text=True, env=env, timeout=300)
    b = subprocess.run([GPS_TEST, path, "2.046e6", "8.184e6", "5000"], capture_output=True, text=True, env=env, timeout=300)
    assert a.returncode == 0 and b.returncode == 0, a.stderr + b.stderr
    assert a.stdout == b.stdout and a.stdout.count("satellite:") == 12 and a.stdout.endswith("run out of file!\n")
