"""GPU: the pipelined front end (gpsacq_pipe_*), the fused 8-bit IQ path (gpsacq_search_iq8, GPSACQ_INPUT=iq_*), the
block decomposition of the single-process multi-GPU entry (gpsacq_multi_search_blocks) and the round-2 advisor items
(sample_spectrum / handoff on a finer Doppler grid).  Everything here is an identity between two routes through the
HIP path, or the HIP path against oracle/iq8_oracle.py, so the comparisons are bit for bit."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def _nott(golden_dir, nblk=64):
    return open(os.path.join(golden_dir, "synth_nott_fs5456.bin"), "rb").read()[:nblk * 5120]


def test_pipe_equals_search(golden_dir):
    """Batches of 1, 1 and a partial third run through three pipeline slots (all in flight before the first collect) give
    the peaks of one plain gpsacq_search of the same blocks, in order."""
    import gpsacq
    buf = np.frombuffer(_nott(golden_dir), dtype=np.uint8)
    with gpsacq.Engine(4.092e6, 5.456e6, 5000.0) as eng:
        _, want = eng.search(buf, want_cells=False)
        batches = [(0, 32), (32, 17), (49, 15)]
        for slot, (b0, n) in enumerate(batches):
            stage = eng.pipe_buffer(slot, 40 * 5120)
            stage[:n * 5120] = buf[b0 * 5120:(b0 + n) * 5120]
            eng.pipe_submit(slot, n)
        got = [eng.pipe_collect(slot) for slot in range(3)]
        # PRN index follows the position in the BATCH (reference schedule of that batch): re-search with explicit tasks
        _, w2 = eng.search(buf, tasks=[(b0 + i, i % 32) for b0, n in batches for i in range(n)], want_cells=False)
        assert np.array_equal(np.concatenate(got), w2)
        assert np.array_equal(got[0], want[:32])
        # a slot cannot be refilled or resubmitted while its search is in flight; an idle one cannot be collected
        stage = eng.pipe_buffer(0, 5120)
        stage[:] = buf[:5120]
        eng.pipe_submit(0, 1)
        with pytest.raises(gpsacq.GpsAcqError):
            eng.pipe_submit(0, 1)
        with pytest.raises(gpsacq.GpsAcqError):
            eng.pipe_buffer(0, 5120)
        assert np.array_equal(eng.pipe_collect(0), want[:1])
        with pytest.raises(gpsacq.GpsAcqError):
            eng.pipe_collect(0)
        with pytest.raises(gpsacq.GpsAcqError):
            eng.pipe_submit(7, 1)


def test_pipe_with_reference_quirk_needs_no_host_wait(golden_dir):
    """ref_quirks on: the cached schedule re-runs only the patch kernel per batch; results equal the synchronous search."""
    import gpsacq
    buf = np.frombuffer(_nott(golden_dir), dtype=np.uint8)
    with gpsacq.Engine(4.092e6, 5.456e6, 5000.0, ref_quirks=True) as eng:
        _, want = eng.search(buf, want_cells=False)
        for slot, b0 in enumerate((0, 32)):
            eng.pipe_buffer(slot, 32 * 5120)[:] = buf[b0 * 5120:(b0 + 32) * 5120]
            eng.pipe_submit(slot, 32)
        got = np.concatenate([eng.pipe_collect(0), eng.pipe_collect(1)])
        assert np.array_equal(got, want)


def _iq_capture(n_blocks, seed=5, fs=2.8e6):
    """Seeded rtl-sdr style capture (uint8 offset 128, baseband): noise + PRN 5 at +1023 Hz + a DC offset."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from make_golden import ca_chips, TAPS, CPS, L1
    rng = np.random.default_rng(seed)
    ns = n_blocks * 40960
    m = np.arange(ns, dtype=np.float64)
    fd = 1023.0
    chips = 1.0 - 2.0 * ca_chips(*TAPS[4])
    idx = np.floor((m + 321) * CPS * (1 + fd / L1) / fs).astype(np.int64) % 1023
    z = (rng.standard_normal(ns) + 1j * rng.standard_normal(ns)) / np.sqrt(2) + 0.3 * chips[idx] * np.exp(2j * np.pi * (fd / fs * m + 0.123))
    z = 30.0 * z + (3.7 + 1.2j)
    iq = np.empty(2 * ns, dtype=np.uint8)
    iq[0::2] = np.clip(np.rint(z.real) + 128, 0, 255).astype(np.uint8)
    iq[1::2] = np.clip(np.rint(z.imag) + 128, 0, 255).astype(np.uint8)
    return iq


@pytest.mark.parametrize("signed,mix,quirks", [(False, 0.62e6, False), (True, 0.0, False), (False, 0.62e6, True)])
def test_fused_iq8_search_equals_convert_then_search(signed, mix, quirks):
    """gpsacq_search_iq8 (conversion fused into the forward transform, no 1-bit intermediate) == gpsacq_iq8_to_bits followed
    by gpsacq_search, cell for cell; the mean matches numpy's and the bits match oracle/iq8_oracle.py."""
    import gpsacq
    from iq8_oracle import iq8_to_bits
    iq = _iq_capture(5, seed=11)
    if signed:
        iq = (iq.astype(np.int16) - 128).astype(np.int8).view(np.uint8)
    with gpsacq.Engine(0.62e6, 2.8e6, 5000.0, ref_quirks=quirks) as eng:
        mean = eng.iq8_mean(iq, signed=signed, chunk_samples=70001)  # ragged pieces
        y = iq.view(np.int8).astype(np.float64) if signed else iq.astype(np.float64) - 128.0
        assert mean == (float(np.mean(y[0::2])), float(np.mean(y[1::2])))
        bits = eng.iq8_to_bits(iq, signed=signed, remove_dc=True, mix_hz=mix, fs=2.8e6)
        ref = iq8_to_bits(iq, signed=signed, remove_dc=True, mix_hz=mix, fs=2.8e6)
        assert np.unpackbits(bits ^ ref).sum() <= 1  # device sincos vs numpy's on a value within an ulp of zero
        tasks = [(b, sv) for b in range(5) for sv in (4, 0, 17)]
        c1, p1 = eng.search(bits, tasks=tasks)
        inp = eng.iq8_input(signed=signed, remove_dc=True, mean=mean, mix_hz=mix, fs=2.8e6, first_sample=0, total_samples=iq.size // 2)
        c2, p2 = eng.search_iq8(iq, inp, tasks=tasks)
        assert np.array_equal(c1, c2) and np.array_equal(p1, p2)
        if mix:  # mixed up to the IF the engine expects, PRN 5 is found (without the mixer the carrier sits at 0 Hz, not at fc)
            assert int(np.argmax([p2["snr"][i] for i in range(0, 3)])) == 0 and p2["snr"][0] > 25
        # a batch that starts in mid-capture: first_sample carries the mixer phase
        off = 2
        inp2 = eng.iq8_input(signed=signed, remove_dc=True, mean=mean, mix_hz=mix, fs=2.8e6, first_sample=off * 40960, total_samples=iq.size // 2)
        c3, _ = eng.search_iq8(iq[off * 81920:], inp2, tasks=[(b - off, sv) for b, sv in tasks if b >= off])
        assert np.array_equal(c3, c1[[i for i, (b, _) in enumerate(tasks) if b >= off]])


def test_multibit_samples_vs_numpy_restatement():
    """gpsacq_iq8_input.multibit (SURVEY.md section 8f.1 "direct float path"): the 8-bit samples keep their amplitude.  Cells
    against oracle/iq8_oracle.py's float64 restatement (LO applied as signs, then the reference's Correlate), and the point of
    the mode: PRN 5's SNR is higher than through the 1-bit path (no quantisation loss)."""
    import gpsacq
    from iq8_oracle import iq8_to_real, multibit_cells, multibit_pwr
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from make_golden import lo_quadrants, code_replica
    iq = _iq_capture(3, seed=21)
    fc, fs = 0.62e6, 2.8e6
    r = r_ = iq8_to_real(iq, remove_dc=True, mix_hz=fc, fs=fs)
    quad = lo_quadrants(fc, fs, 40960)
    with gpsacq.Engine(fc, fs, 5000.0) as eng:
        mean = eng.iq8_mean(iq)
        tasks = [(0, 4), (2, 4), (1, 0)]
        inp = eng.iq8_input(remove_dc=True, mean=mean, mix_hz=fc, fs=fs, total_samples=iq.size // 2, multibit=True)
        cells, peaks = eng.search_iq8(iq, inp, tasks=tasks)
        inp1 = eng.iq8_input(remove_dc=True, mean=mean, mix_hz=fc, fs=fs, total_samples=iq.size // 2)
        _, peaks1 = eng.search_iq8(iq, inp1, tasks=tasks)
        for t, (b, sv) in enumerate(tasks):
            mp, mi, tp = multibit_cells(r[b * 40960:], quad, code_replica(fs, sv), eng.dmax, eng.num_lags)
            np.testing.assert_allclose(cells["max_pwr"][t], mp, rtol=2e-5)
            np.testing.assert_allclose(cells["tot_pwr"][t], tp, rtol=2e-5)
            assert (cells["max_i"][t] != mi).sum() <= 1
        assert peaks["snr"][0] > 25 and peaks["lo_shift"][0] == peaks1["lo_shift"][0] and peaks["ca_shift"][0] == peaks1["ca_shift"][0]
        assert peaks["snr"][0] > 1.1 * peaks1["snr"][0] and peaks["snr"][1] > 1.1 * peaks1["snr"][1]
        # a Doppler grid finer than a bin (70 Hz bins, 30 Hz asked -> 3 sub-bin copies of the float samples, 23.3 Hz): points against the
        # restatement with the sub-bin turn of oracle_sample_ramped applied to the samples; whole-bin points equal the bin grid's cells
        eng.set_doppler_step(30.0)
        assert eng.doppler_sub == 3
        K = eng.kmax
        cg, pg = eng.search_iq8(iq, inp, tasks=tasks)
        assert cg.shape == (3, 2 * K + 1)
        ks = np.arange(-K, K + 1)
        for t in range(3):
            assert np.array_equal(cg[t][ks % 3 == 0], cells[t][ks[ks % 3 == 0] // 3 + eng.dmax])
        for t, (b, sv) in enumerate(tasks[:2]):
            for r in (1, 2):
                pts = [k for k in (-K, -K + 1, -7, -5, -2, -1, 1, 2, 4, 8, K - 1, K) if k % 3 == r]
                mp, mi, tp = multibit_cells(r_[b * 40960:], quad, code_replica(fs, sv), eng.dmax, eng.num_lags, eps=r / 3.0, dops=[(k - r) // 3 for k in pts])
                got = cg[t][np.array(pts) + K]
                np.testing.assert_allclose(got["max_pwr"], mp, rtol=2e-5)
                np.testing.assert_allclose(got["tot_pwr"], tp, rtol=2e-5)
                assert (got["max_i"] != mi).sum() <= 1
            assert pg["snr"][t] >= peaks["snr"][t] and abs(pg["lo_shift"][t] / 3.0 - peaks["lo_shift"][t]) <= 1.0 and pg["ca_shift"][t] == peaks["ca_shift"][t]
        eng.set_doppler_step(0.0)
        # non-coherent accumulation over the float samples: two blocks' powers summed (plain sum: creep re-alignment off)
        eng.set_noncoherent(2, 1)
        eng.set_creep_compensation(False)
        ca, pa = eng.search_iq8(iq, inp, tasks=[(0, 4)])
        acc = [multibit_pwr(r_[b * 40960:], quad, code_replica(fs, 4), eng.dmax, eng.num_lags) for b in (0, 1)]
        pw = acc[0] + acc[1]
        np.testing.assert_allclose(ca["max_pwr"][0], pw.max(axis=1), rtol=2e-5)
        np.testing.assert_allclose(ca["tot_pwr"][0], pw.sum(axis=1), rtol=2e-5)
        assert (ca["max_i"][0] != pw.argmax(axis=1)).sum() <= 1  # (the blocks are 40960 samples apart, not whole code periods: parity only)
        # ... and with gpsacq_set_block_alignment: block 1 moved back by 40960 mod 2800 = 1760 samples of code phase -- PRN 5 adds up
        eng.set_block_alignment(True)
        cb, pb = eng.search_iq8(iq, inp, tasks=[(0, 4)])
        pw = acc[0] + np.roll(acc[1], -(40960 % 2800), axis=1)
        np.testing.assert_allclose(cb["max_pwr"][0], pw.max(axis=1), rtol=2e-5)
        np.testing.assert_allclose(cb["tot_pwr"][0], pw.sum(axis=1), rtol=2e-5)
        assert (cb["max_i"][0] != pw.argmax(axis=1)).sum() <= 1
        assert pb["snr"][0] > 25 and pb["snr"][0] > pa["snr"][0] and pb["ca_shift"][0] == peaks["ca_shift"][0]
        # the sign path of the same capture through the same switch equals the search of the converted bits
        inp_sign = eng.iq8_input(remove_dc=True, mean=mean, mix_hz=fc, fs=fs, total_samples=iq.size // 2)
        c_iq, _ = eng.search_iq8(iq, inp_sign, tasks=[(0, 4)])
        c_bits, _ = eng.search(eng.iq8_to_bits(iq, remove_dc=True, mix_hz=fc, fs=fs), tasks=[(0, 4)])
        assert np.array_equal(c_iq, c_bits)
        eng.set_block_alignment(False)
        eng.set_noncoherent(1, 1)
    # not with the reference quirk
    with gpsacq.Engine(fc, fs, 5000.0, ref_quirks=True) as eng:
        with pytest.raises(gpsacq.GpsAcqError):
            eng.search_iq8(iq, inp, tasks=tasks)


def test_cli_iq8_input_equals_preconverted_1bit(tmp_path):
    """README.md:83-115 as one command: gps_test on the rtl-sdr IQ file (GPSACQ_INPUT=iq_u8, GPSACQ_MIX_HZ) prints what
    gps_test prints on the 1-bit file made from it first -- two runs, the partial third discarded by both."""
    import gpsacq
    from test_host import GPS_TEST, BANNER
    iq = _iq_capture(70, seed=3)
    with gpsacq.Engine(0.62e6, 2.8e6, 5000.0) as eng:
        bits = eng.iq8_to_bits(iq, remove_dc=True, mix_hz=0.62e6, fs=2.8e6)
    f_iq, f_bits = str(tmp_path / "cap_iq.bin"), str(tmp_path / "cap_1bit.bin")
    iq.tofile(f_iq)
    bits.tofile(f_bits)
    args = ["0.62e6", "2.8e6", "5000"]
    for quirks in ("0", "1"):
        env = dict(os.environ, GPSACQ_REF_QUIRKS=quirks, GPSACQ_BATCH_RUNS="1")
        a = subprocess.run([GPS_TEST, f_bits] + args, capture_output=True, text=True, env=env, timeout=300)
        b = subprocess.run([GPS_TEST, f_iq] + args, capture_output=True, text=True, timeout=300,
                           env=dict(env, GPSACQ_INPUT="iq_u8", GPSACQ_MIX_HZ="0.62e6", GPSACQ_TRACE="1"))
        assert a.returncode == 0 and b.returncode == 0, a.stderr + b.stderr
        assert a.stdout == b.stdout
        assert a.stdout.startswith(BANNER) and a.stdout.count("satellite:") == 2 and a.stdout.endswith("run out of file!\n")
        # the capture's mean from the host threads (default) and from the GPU pass are the same integers
        c = subprocess.run([GPS_TEST, f_iq] + args, capture_output=True, text=True, timeout=300,
                           env=dict(env, GPSACQ_INPUT="iq_u8", GPSACQ_MIX_HZ="0.62e6", GPSACQ_SUMS_ON_GPU="1"))
        d = subprocess.run([GPS_TEST, f_iq] + args, capture_output=True, text=True, timeout=300,
                           env=dict(env, GPSACQ_INPUT="iq_u8", GPSACQ_MIX_HZ="0.62e6", GPSACQ_SUM_THREADS="3"))
        assert c.returncode == 0 and d.returncode == 0 and c.stdout == b.stdout and d.stdout == b.stdout
        # ... and the capture read with fread only (no mapping) gives the same report, IQ and 1-bit
        for f_in, extra, ref in ((f_iq, dict(GPSACQ_INPUT="iq_u8", GPSACQ_MIX_HZ="0.62e6"), b), (f_bits, {}, a)):
            n = subprocess.run([GPS_TEST, f_in] + args, capture_output=True, text=True, timeout=300, env=dict(env, GPSACQ_NO_MMAP="1", **extra))
            assert n.returncode == 0 and n.stdout == ref.stdout
        assert "gpsacq trace" in b.stderr and "input iq_u8" in b.stderr
    # PRN 5 (index 4) is a hit in both runs
    lines = a.stdout[len(BANNER):].split("\n")
    assert "satellite" in lines[0] and "    4 " in lines[0]
    r = subprocess.run([GPS_TEST, f_iq] + args, capture_output=True, text=True, env=dict(os.environ, GPSACQ_INPUT="wav"), timeout=120)
    assert r.returncode != 0 and "GPSACQ_INPUT" in r.stderr
    # HackRF format (int8, proc_hackrf_bin_for_gps.m:7-19: the real part, no mixer) and the multi-bit switch
    s8 = (iq.astype(np.int16) - 128).astype(np.int8)
    f_s8, f_b8 = str(tmp_path / "cap_s8.bin"), str(tmp_path / "cap_s8_1bit.bin")
    s8.tofile(f_s8)
    with gpsacq.Engine(0.62e6, 2.8e6, 5000.0) as eng:
        eng.iq8_to_bits(s8, signed=True, remove_dc=True, mix_hz=0.0, fs=2.8e6).tofile(f_b8)
    a = subprocess.run([GPS_TEST, f_b8] + args, capture_output=True, text=True, timeout=300)
    b = subprocess.run([GPS_TEST, f_s8] + args, capture_output=True, text=True, timeout=300, env=dict(os.environ, GPSACQ_INPUT="iq_s8"))
    assert a.returncode == 0 and b.returncode == 0 and a.stdout == b.stdout and a.stdout.count("satellite:") == 2
    m = subprocess.run([GPS_TEST, f_iq] + args, capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, GPSACQ_INPUT="iq_u8", GPSACQ_MIX_HZ="0.62e6", GPSACQ_IQ_MULTIBIT="1"))
    assert m.returncode == 0 and m.stdout.count("satellite:") == 2 and "    4 " in m.stdout[len(BANNER):].split("\n")[0]


def test_cli_searches_the_converters_hackrf_file_like_the_1bit_capture(golden_dir, tmp_path):
    """The reference has two files for one capture: the 1-bit stream gps_test reads and the int8 IQ file
    c/conv_1bit_bin_to_hackrf_bin.cpp makes of it for HackRF replay (already mixed to baseband by the same XOR LO).
    GPSACQ_INPUT=iq_s8 GPSACQ_IQ_COMPLEX=1 searches the second directly: same report as the first, all 12 runs."""
    from test_host import GPS_TEST
    from oracle_lib import lib, _p
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from iq8_oracle import hackrf_replay_file
    path = os.path.join(golden_dir, "gps_sig_tmp.bin")
    fc, fs = 2.046e6, 8.184e6
    bits = np.fromfile(path, dtype=np.uint8)
    quad = np.zeros(bits.size * 8, np.uint8)
    lib().oracle_lo_quadrants(fc, fs, quad.size, _p(quad))
    f_iq = str(tmp_path / "replay_s8.bin")
    hackrf_replay_file(bits, quad).tofile(f_iq)
    args = ["2.046e6", "8.184e6", "5000"]
    a = subprocess.run([GPS_TEST, path] + args, capture_output=True, text=True, timeout=300)
    b = subprocess.run([GPS_TEST, f_iq] + args, capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, GPSACQ_INPUT="iq_s8", GPSACQ_IQ_COMPLEX="1", GPSACQ_IQ_KEEP_DC="1"))
    assert a.returncode == 0 and b.returncode == 0, a.stderr + b.stderr
    assert a.stdout.count("satellite:") == 12 and a.stdout == b.stdout


def test_cli_streams_batches_of_growing_size(golden_dir):
    """The front end's output does not depend on how the file is cut into batches (1, 2, 4 ... runs; capped)."""
    from test_host import GPS_TEST
    path = os.path.join(golden_dir, "gps_sig_tmp.bin")
    outs = []
    for batch in ("1", "3", "64"):
        r = subprocess.run([GPS_TEST, path, "2.046e6", "8.184e6", "5000"], capture_output=True, text=True, timeout=300,
                           env=dict(os.environ, GPSACQ_BATCH_RUNS=batch))
        assert r.returncode == 0, r.stderr
        outs.append(r.stdout)
    assert outs[0] == outs[1] == outs[2] and outs[0].count("satellite:") == 12


def test_multi_search_blocks_single_device(golden_dir):
    """gpsacq_multi_search_blocks in its one-device case: every (run, PRN) peak equals gpsacq_search's, and best[] is the
    per-PRN maximum under the key order (higher SNR; ties to the lower Doppler point)."""
    import gpsacq
    buf = _nott(golden_dir)
    with gpsacq.Engine(4.092e6, 5.456e6, 5000.0) as eng:
        _, want = eng.search(buf, want_cells=False)
        kmax = eng.kmax
    with gpsacq.MultiEngine(4.092e6, 5.456e6, 5000.0, devices=(0,)) as me:
        peaks, best = me.search_blocks(buf)
        assert np.array_equal(peaks, want)
        for sv in range(32):
            cand = want[sv::32]
            keys = [(float(p["snr"]), -int(p["lo_shift"]), int(p["ca_shift"])) for p in cand]
            k = max(keys)
            assert (float(best["snr"][sv]), -int(best["lo_shift"][sv]), int(best["ca_shift"][sv])) == k
        # the grid entry point still sees the whole Doppler range afterwards (windows are put back)
        tasks = [(0, sv) for sv in range(32)]
        pk = me.search_grid(buf, tasks)
        with gpsacq.Engine(4.092e6, 5.456e6, 5000.0) as eng:
            _, w = eng.search(buf, tasks=tasks, want_cells=False)
        assert np.array_equal(pk, w)  # max_pwr too: the winner's value is merged next to the keys
        assert me.kmax == kmax
        with pytest.raises(ValueError):
            me.search_blocks(buf[:31 * 5120])


@pytest.mark.parametrize("n_eng", [2, 3, 5])
def test_multi_engines_sharing_one_gpu(golden_dir, n_eng):
    """The N > 1 control flow of gpsacq_multi_* on the one-GPU box: N engines on device 0 (their keys merged on the device;
    RCCL only joins distinct GPUs).  Block decomposition: every (run, PRN) peak and the per-PRN best equal the one-engine
    results, max_pwr included; grid decomposition (Doppler slabs): the merged peaks equal the unsharded search."""
    import gpsacq
    buf = _nott(golden_dir)
    with gpsacq.Engine(4.092e6, 5.456e6, 5000.0) as eng:
        _, want = eng.search(buf, want_cells=False)
        tasks = [(b, sv) for b in (0, 33) for sv in range(32)]
        _, wgrid = eng.search(buf, tasks=tasks, want_cells=False)
    with gpsacq.MultiEngine(4.092e6, 5.456e6, 5000.0, devices=(0,) * n_eng) as me:
        assert me.n_devices == n_eng
        for _ in range(2):  # twice: buffers are reused
            peaks, best = me.search_blocks(buf)  # 2 runs over n_eng engines: some engines get no run at all
            assert np.array_equal(peaks, want)
            for sv in range(32):
                cand = want[sv::32]
                k = max((float(p["snr"]), -int(p["lo_shift"]), int(p["ca_shift"])) for p in cand)
                assert (float(best["snr"][sv]), -int(best["lo_shift"][sv]), int(best["ca_shift"][sv])) == k
                w = [p for p in cand if (float(p["snr"]), -int(p["lo_shift"]), int(p["ca_shift"])) == k][0]
                assert float(best["max_pwr"][sv]) == float(w["max_pwr"])
            got = me.search_grid(buf, tasks)
            assert np.array_equal(got, wgrid)  # snr, lo_shift, ca_shift and the winner's max_pwr


def test_sample_spectrum_on_a_fresh_fine_grid_engine(golden_dir):
    """Advisor r2: gpsacq_sample_spectrum after gpsacq_set_doppler_step on an engine that never searched used to write
    `sub` spectra into a one-spectrum buffer.  The probe now transforms exactly one (offset 0) spectrum."""
    import gpsacq
    blk = _nott(golden_dir, 1)
    with gpsacq.Engine(4.092e6, 5.456e6, 5000.0) as a, gpsacq.Engine(4.092e6, 5.456e6, 5000.0) as b:
        want = a.sample_spectrum(blk)
        b.set_doppler_step(50.0)
        assert b.doppler_sub == 3
        got = b.sample_spectrum(blk)
        assert np.array_equal(got, want)
        cells, peaks = b.search(blk, tasks=[(0, 0)])  # and the engine still searches
        assert cells.shape == (1, b.num_doppler)
        # the reference grid comes back exactly
        b.set_doppler_step(0.0)
        assert (b.num_doppler, b.first_doppler, b.doppler_sub) == (a.num_doppler, a.first_doppler, 1)


def test_handoff_follows_the_doppler_grid(golden_dir):
    """Advisor r2: lo_shift counts grid points after gpsacq_set_doppler_step; the hand-off must use that step."""
    import gpsacq
    buf = _nott(golden_dir, 1)
    with gpsacq.Engine(4.092e6, 5.456e6, 5000.0) as eng:
        _, p_bin = eng.search(buf, tasks=[(0, 0)], want_cells=False)
        h_bin = eng.handoff(p_bin[0])
        assert h_bin == gpsacq.handoff(p_bin[0], 4.092e6, 5.456e6)
        eng.set_doppler_step(50.0)
        _, p_fine = eng.search(buf, tasks=[(0, 0)], want_cells=False)
        h_fine = eng.handoff(p_fine[0])
        assert abs(h_fine["lo_dop_hz"] - int(p_fine["lo_shift"][0]) * eng.doppler_step_hz) < 1e-9
        assert abs(h_fine["lo_dop_hz"] - h_bin["lo_dop_hz"]) <= 5.456e6 / 40000  # the same satellite, within a bin
        assert h_fine == gpsacq.handoff(p_fine[0], 4.092e6, 5.456e6, step_hz=eng.doppler_step_hz)
        # read as FFT bins (the old behaviour) the fine-grid index is three times too far out
        wrong = gpsacq.handoff(p_fine[0], 4.092e6, 5.456e6)
        if int(p_fine["lo_shift"][0]) != 0:
            assert abs(wrong["lo_dop_hz"]) > 2.5 * abs(h_fine["lo_dop_hz"])
