"""Non-coherent accumulation (extension, SURVEY.md section 8f.2 / BASELINE config 4): the device path
against the oracle's restatement (sum of per-lag powers over blocks, then the reference's scan)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_noncoherent_vs_oracle(golden_dir):
    import gpsacq
    from oracle_lib import Oracle
    buf = open(os.path.join(golden_dir, "synth_weak_fs5456.bin"), "rb").read()
    orc = Oracle(4.092e6, 5.456e6, 5000.0)
    with gpsacq.Engine(4.092e6, 5.456e6, 5000.0) as eng:
        stride = eng.aligned_stride()
        assert stride == 5456  # 8 C/A periods of 682 bytes
        tasks = [(0, 11), (0, 2), (0, 20), (1, 11)]
        coh_cells, coh_peaks = eng.search(buf, tasks=tasks, stride=stride)
        eng.set_noncoherent(5, 1)
        nc_cells, nc_peaks = eng.search(buf, tasks=tasks, stride=stride)
        for t, (b, sv) in enumerate(tasks):
            want = orc.search_noncoherent(buf, stride, b, sv, 5, 1)
            np.testing.assert_allclose(nc_cells["max_pwr"][t], want["max_pwr"], rtol=2e-5)
            np.testing.assert_allclose(nc_cells["tot_pwr"][t], want["tot_pwr"], rtol=2e-5)
            assert (nc_cells["max_i"][t] != want["max_i"]).sum() <= 1
        # the point of the mode: PRN 12 (weak) is buried coherently, stands out non-coherently
        noise_coh, noise_nc = coh_peaks["snr"][2], nc_peaks["snr"][2]
        assert coh_peaks["snr"][0] < 1.3 * noise_coh
        assert nc_peaks["snr"][0] > 2.0 * noise_nc
        assert int(nc_peaks["lo_shift"][0]) == 4 and abs(int(nc_peaks["ca_shift"][0]) - 1000) <= 1
        assert int(nc_peaks["lo_shift"][1]) == -11 and int(nc_peaks["ca_shift"][1]) == 3333
        # out-of-range accumulation span is rejected, n_acc = 1 restores the coherent results bit for bit
        with pytest.raises(gpsacq.GpsAcqError):
            eng.search(buf, tasks=[(3, 0)], stride=stride)
        eng.set_noncoherent(1)
        again_cells, again_peaks = eng.search(buf, tasks=tasks, stride=stride)
        assert np.array_equal(again_cells, coh_cells) and np.array_equal(again_peaks, coh_peaks)
        # default schedule with accumulation over a stride of blocks: task t = (block t, prn t % 32)
        eng.set_noncoherent(3, 2)
        c, p = eng.search(buf, stride=stride, n_tasks=3)
        want = orc.search_noncoherent(buf, stride, 2, 2, 3, 2)
        np.testing.assert_allclose(c["max_pwr"][2], want["max_pwr"], rtol=2e-5)


def test_bench_size_batch_properties():
    """At bench.py's batch size the oracle is out of reach; check size-independent properties:
    a batch equals its halves, a permutation of the blocks permutes the results, peaks equal the
    reference's Doppler scan (:196-198) over the cells, and reruns are bit-identical."""
    import gpsacq
    rng = np.random.default_rng(11)
    nblk = 2048
    bits = rng.integers(0, 256, nblk * 5120, dtype=np.uint8)
    with gpsacq.Engine(4.092e6, 5.456e6, 5000.0) as eng:
        cells, peaks = eng.search(bits)
        c1, p1 = eng.search(bits[:1024 * 5120])
        c2, p2 = eng.search(bits[1024 * 5120:])  # blocks 1024.. keep their PRN (1024 % 32 == 0)
        assert np.array_equal(cells[:1024], c1) and np.array_equal(cells[1024:], c2)
        assert np.array_equal(peaks[:1024], p1) and np.array_equal(peaks[1024:], p2)
        perm = rng.permutation(nblk)
        tasks = np.stack([perm, perm % 32], axis=1)
        cp, pp = eng.search(bits, tasks=tasks)
        assert np.array_equal(cp, cells[perm]) and np.array_equal(pp, peaks[perm])
        # k_peaks against a host restatement of the scan
        best = np.zeros(nblk, dtype=np.int64)
        snr = cells["snr"]
        for t in range(nblk):
            best[t] = int(np.argmax(snr[t]))  # first maximum == strict '>' scanning upwards
        assert np.array_equal(peaks["lo_shift"], best - eng.dmax)
        assert np.array_equal(peaks["ca_shift"], cells["max_i"][np.arange(nblk), best])
        assert np.array_equal(peaks["snr"], snr[np.arange(nblk), best])
        assert np.isfinite(snr).all() and (cells["max_i"] >= 0).all() and (cells["max_i"] < eng.num_lags).all()
        # noise-only capture: no cell should look like a satellite
        assert peaks["snr"].max() < 25


def test_doppler_slabs_merge_to_full_search(golden_dir):
    """Multi-GPU decomposition by Doppler slab, emulated on one GPU: two windows searched separately
    and merged with the packed-key MAX (gpsacq.dist) must equal the full-range search."""
    import torch
    import gpsacq
    from gpsacq import dist as D
    buf = open(os.path.join(golden_dir, "synth_nott_fs5456.bin"), "rb").read()[:5 * 5120]
    tasks = [(b, sv) for b in (0, 4) for sv in range(32)]
    with gpsacq.Engine(4.092e6, 5.456e6, 20000.0) as eng:
        full_cells, full_peaks = eng.search(buf, tasks=tasks)
        merged = None
        for rank in range(3):
            first, n = D.shard_doppler(eng.dmax, rank, 3)
            eng.set_doppler_window(first, n)
            cells, peaks = eng.search(buf, tasks=tasks)
            assert cells.shape[1] == n
            assert np.array_equal(cells, full_cells[:, first + eng.dmax:first + eng.dmax + n])
            key = D.pack_keys(torch.from_numpy(peaks.view(np.int32).reshape(-1, 4).copy()), eng.dmax)
            merged = key if merged is None else torch.maximum(merged, key)
        snr, lo, ca = D.unpack_keys(merged, eng.dmax)
        assert np.array_equal(snr.numpy(), full_peaks["snr"])
        assert np.array_equal(lo.numpy(), full_peaks["lo_shift"]) and np.array_equal(ca.numpy(), full_peaks["ca_shift"])
        with pytest.raises(gpsacq.GpsAcqError):
            eng.set_doppler_window(-eng.dmax - 1, 3)


def test_forward_chunking_beyond_8192_blocks():
    """More blocks than one forward-transform launch covers (8192): the chunk seam must be invisible."""
    import gpsacq
    rng = np.random.default_rng(5)
    nblk = 8192 + 40
    bits = rng.integers(0, 256, nblk * 5120, dtype=np.uint8)
    with gpsacq.Engine(2.046e6, 8.184e6, 5000.0) as eng:  # the 33-column kernel instance
        tasks = np.array([(b, b % 32) for b in list(range(8180, nblk)) + [0, 1, 4095]], dtype=np.int32)
        cells, peaks = eng.search(bits, tasks=tasks)
        tail = bits[8180 * 5120:]
        c2, p2 = eng.search(tail, tasks=np.array([(b - 8180, b % 32) for b in range(8180, nblk)], dtype=np.int32))
        assert np.array_equal(cells[:nblk - 8180], c2) and np.array_equal(peaks[:nblk - 8180], p2)
        c3, p3 = eng.search(bits[:4096 * 5120], tasks=np.array([(0, 0), (1, 1), (4095, 4095 % 32)], dtype=np.int32))
        assert np.array_equal(cells[nblk - 8180:], c3)


def test_configs3_weak_signal_rtl_path(golden_dir):
    """BASELINE configs[3] ("weak-signal ... +-100 kHz ... rtl_sdr 2.8 Msps path") as built here:
    fs 2.8 MHz, IF 0.62 MHz, max_fo 100 kHz honoured (2857 bins of 70 Hz, N = 40000 = 14.3 ms
    coherent), 5 non-coherent sums over blocks 15 C/A periods apart."""
    import gpsacq
    from oracle_lib import Oracle
    buf = open(os.path.join(golden_dir, "synth_weak_rtl_fs2800.bin"), "rb").read()
    with gpsacq.Engine(0.62e6, 2.8e6, 100000.0) as eng:
        assert eng.num_doppler == 2857 and eng.num_lags == 2800 and eng.aligned_stride() == 5250
        eng.set_noncoherent(5, 1)
        cells, peaks = eng.search(buf, tasks=[(0, 8), (0, 30)], stride=5250)  # PRN 9 present, PRN 31 absent
        assert int(peaks["lo_shift"][0]) == 571 and abs(int(peaks["ca_shift"][0]) - 777) <= 1
        assert peaks["snr"][0] > 1.8 * peaks["snr"][1]
        # a window of the grid against the oracle's restatement
        orc = Oracle(0.62e6, 2.8e6, 100000.0)
        want = orc.search_noncoherent(buf, 5250, 0, 8, 5, 1, first_bin=540, n_bins=60)
        got = cells[0][540 + eng.dmax:600 + eng.dmax]
        np.testing.assert_allclose(got["max_pwr"], want["max_pwr"], rtol=2e-5)
        np.testing.assert_allclose(got["tot_pwr"], want["tot_pwr"], rtol=2e-5)
        assert (got["max_i"] != want["max_i"]).sum() <= 1


@pytest.mark.parametrize("fc,fs,max_fo", [(2.6e6, 10e6, 5000.0), (1.023e6, 4.092e6, 3000.0), (3.5e6, 9.9987e6, 5000.0),
                                          (0.0, 1.1e6, 1000.0), (4.092e6, 16.368e6, 4000.0), (9.5e6, 38.192e6, 2000.0),
                                          (1.7e6, 6.8e6, 5000.0)])
def test_other_sampling_rates(fc, fs, max_fo):
    """Every kernel instance (12/22/28/33/40 accumulator columns), lag counts that are not multiples of
    250 or of 8, a zero IF, and more than 10000 lags (fs > 10 MHz: 2 and 4 passes of 40 columns plus
    the merge kernel): cells against the oracle on a seeded noise + signal capture."""
    import gpsacq
    from oracle_lib import Oracle
    rng = np.random.default_rng(int(fs) % 1000 + 17)
    ns = 3 * 40960
    m = np.arange(ns, dtype=np.float64)
    chips = np.where(rng.integers(0, 2, 1023) == 1, -1.0, 1.0)  # any +-1 sequence will do as interference
    y = rng.standard_normal(ns) + 0.2 * chips[np.floor(m * 1.023e6 / fs).astype(np.int64) % 1023] * np.cos(2 * np.pi * (fc + 700.0) / fs * m)
    bits = np.packbits((y < 0).astype(np.uint8), bitorder="little").tobytes()
    orc = Oracle(fc, fs, max_fo)
    with gpsacq.Engine(fc, fs, max_fo) as eng:
        assert eng.dmax == orc.dmax and eng.num_lags == orc.num_lags
        tasks = [(0, 0), (1, 17), (2, 31)]
        cells, peaks = eng.search(bits, tasks=tasks)
        ocells, opeaks = orc.search(bits, tasks)
        np.testing.assert_allclose(cells["max_pwr"], ocells["max_pwr"], rtol=2e-5)
        np.testing.assert_allclose(cells["tot_pwr"], ocells["tot_pwr"], rtol=2e-5)
        assert (cells["max_i"] != ocells["max_i"]).mean() < 0.01
        for t in range(len(tasks)):
            if peaks["lo_shift"][t] == opeaks["lo_shift"][t]:
                assert peaks["ca_shift"][t] == opeaks["ca_shift"][t]
                continue
            # another Doppler bin may only win where the oracle's own two SNRs tie to rounding (a zero IF makes bins +d and -d
            # mirror images of each other: equal powers, and the strict '>' scan keeps whichever rounds higher)
            k_gpu, k_orc = int(peaks["lo_shift"][t]) + orc.dmax, int(opeaks["lo_shift"][t]) + orc.dmax
            assert abs(ocells["snr"][t][k_gpu] / ocells["snr"][t][k_orc] - 1) < 2e-5, (t, peaks[t], opeaks[t])
            assert peaks["ca_shift"][t] == ocells["max_i"][t][k_gpu]


def test_cli_multi_engine_threads(golden_dir):
    """GPSACQ_DEVICES splits each batch of runs over one host thread + engine per listed device.  With
    one GPU the list '0,0,0' exercises exactly that code path; stdout must not change."""
    import subprocess
    from test_host import GPS_TEST
    path = os.path.join(golden_dir, "gps_sig_tmp.bin")
    base = dict(os.environ, GPSACQ_BATCH_RUNS="2")
    one = subprocess.run([GPS_TEST, path, "2.046e6", "8.184e6", "5000"], capture_output=True, text=True, env=base, timeout=300)
    many = subprocess.run([GPS_TEST, path, "2.046e6", "8.184e6", "5000"], capture_output=True, text=True,
                          env=dict(base, GPSACQ_DEVICES="0,0,0"), timeout=300)
    assert one.returncode == 0 and many.returncode == 0, many.stderr
    assert one.stdout == many.stdout and one.stdout.count("satellite:") == 12
    bad = subprocess.run([GPS_TEST, path, "2.046e6", "8.184e6", "5000"], capture_output=True, text=True,
                         env=dict(base, GPSACQ_DEVICES="0,99"), timeout=300)
    assert bad.returncode == 1 and "SearchInit() returned 1" in bad.stdout


def test_device_generated_capture_is_found():
    """Synthetic capture made on the device (gps_sig_gen.m's role): the injected PRNs
    must come back from the search with the Doppler bin and code phase they were generated with."""
    import gpsacq
    fs, fc = 5.456e6, 4.092e6
    sats = [(1, 0.151, 6 * fs / 40000, 1465.0, 0.1), (21, 0.151, 8 * fs / 40000, 686.0, 0.7),
            (29, 0.2, -9 * fs / 40000, 3868.0, 0.3), (5, 0.12, 2501.0, 12.25, 0.0)]
    with gpsacq.Engine(fc, fs, 5000.0) as eng:
        bits = eng.generate(4 * 5120, sats, noise_sigma=1.0, seed=42)
        again = eng.generate(4 * 5120, sats, noise_sigma=1.0, seed=42)
        other = eng.generate(4 * 5120, sats, noise_sigma=1.0, seed=43)
        assert np.array_equal(bits, again) and not np.array_equal(bits, other)
        ones = np.unpackbits(bits).mean()
        assert 0.48 < ones < 0.52
        tasks = [(b, sv) for b in range(4) for sv in range(32)]
        _, peaks = eng.search(bits, tasks=tasks)
        peaks = peaks.reshape(4, 32)
        for prn, amp, dop, ca, ph in sats:
            for b in range(4):
                pk = peaks[b, prn - 1]
                assert pk["snr"] > 30, (prn, b, pk)
                assert abs(int(pk["lo_shift"]) - round(dop * 40000 / fs)) <= 1
                expect = (ca + 40960 * b * (1 + dop / 1575.42e6)) % 5456
                d = abs(int(pk["ca_shift"]) - expect)
                assert min(d, 5456 - d) <= 1.5, (prn, b, pk, expect)
        absent = [sv for sv in range(32) if sv + 1 not in (1, 21, 29, 5)]
        assert peaks[:, absent]["snr"].max() < 25
        noise_only = eng.generate(5120, (), seed=7)
        _, p0 = eng.search(noise_only, tasks=[(0, sv) for sv in range(32)])
        assert p0["snr"].max() < 25


def test_device_entry_point_and_task_bounds():
    """gpsacq_search_device (buffers already in HBM, engine stream): same answers as the host-pointer
    entry point; a device task list is bounded by the kernel -- out-of-range tasks come back as empty
    cells (max_i = -1, snr = 0) and the valid ones are untouched."""
    import torch
    import gpsacq
    fs, fc = 5.456e6, 4.092e6
    with gpsacq.Engine(fc, fs, 5000.0) as eng:
        bits = eng.generate(3 * 5120, [(9, 0.2, 1000.0, 777.0, 0.0)], seed=5)
        tasks = np.array([(0, 8), (2, 8), (3, 8), (1, 32), (-1, 0), (1, 8), (0, -5)], dtype=gpsacq.TASK_DTYPE)
        valid = [0, 1, 5]
        cells_h, peaks_h = eng.search(bits, tasks=[tuple(t) for t in tasks[valid]], want_cells=True)
        d_bits = torch.from_numpy(bits).cuda()
        d_tasks = torch.from_numpy(tasks.view(np.int32).reshape(-1, 2)).cuda()
        d_cells = torch.zeros(len(tasks) * eng.num_doppler * 4, dtype=torch.int32, device="cuda")
        d_peaks = torch.zeros(len(tasks) * 4, dtype=torch.int32, device="cuda")
        eng.search_device(d_bits.data_ptr(), 3, d_peaks.data_ptr(), d_tasks_ptr=d_tasks.data_ptr(), n_tasks=len(tasks),
                          d_cells_ptr=d_cells.data_ptr(), sync=True)
        cells_d = d_cells.cpu().numpy().view(gpsacq.CELL_DTYPE).reshape(len(tasks), eng.num_doppler)
        peaks_d = d_peaks.cpu().numpy().view(gpsacq.PEAK_DTYPE)
        assert np.array_equal(cells_d[valid], cells_h)
        assert np.array_equal(peaks_d[valid], peaks_h)
        assert peaks_h["snr"].min() > 30
        for t in (2, 3, 4, 6):
            assert (cells_d[t]["max_i"] == -1).all() and (cells_d[t]["snr"] == 0).all()
            assert peaks_d[t]["snr"] == 0
        # host task lists are still rejected up front
        with pytest.raises(gpsacq.GpsAcqError):
            eng.search(bits, tasks=[(3, 8)])


def test_async_searches_timing_ring_and_stream():
    """Searches enqueued with sync=0 on the engine's stream: results read on another stream after an
    event wait equal the synchronous ones; gpsacq_timing_ago() keeps the last 8 searches apart."""
    import torch
    import gpsacq
    with gpsacq.Engine(4.092e6, 5.456e6, 5000.0) as eng:
        with pytest.raises(gpsacq.GpsAcqError):
            eng.last_timing()  # nothing has run yet
        n = 64
        bits = eng.generate(n * 5120, [(3, 0.2, -1500.0, 100.0, 0.0)], seed=11)
        _, want = eng.search(bits, want_cells=False)
        assert eng.last_timing()["cells"] == n * eng.num_doppler
        d_bits = torch.from_numpy(bits).cuda()
        stream = torch.cuda.ExternalStream(eng.stream_ptr)
        assert eng.stream_ptr != 0
        bufs = [torch.zeros(n * 4, dtype=torch.int32, device="cuda") for _ in range(3)]
        half = torch.zeros((n // 2) * 4, dtype=torch.int32, device="cuda")
        copies = []
        for b in bufs:
            eng.search_device(d_bits.data_ptr(), n, b.data_ptr(), sync=False)
            ev = torch.cuda.Event()
            ev.record(stream)
            torch.cuda.current_stream().wait_event(ev)
            copies.append(b.clone())
        eng.search_device(d_bits.data_ptr(), n // 2, half.data_ptr(), sync=False)
        t0, t1 = eng.last_timing(0), eng.last_timing(1)
        assert t0["cells"] == (n // 2) * eng.num_doppler and t1["cells"] == n * eng.num_doppler
        assert 0 < t0["ms_correlate"] < t1["ms_correlate"] * 1.5 and t1["ms_total"] >= t1["ms_correlate"]
        with pytest.raises(gpsacq.GpsAcqError):
            eng.last_timing(8)
        with pytest.raises(gpsacq.GpsAcqError):
            eng.last_timing(5)  # only 5 searches so far (indices 0..4)
        torch.cuda.synchronize()
        for c in copies:
            assert np.array_equal(c.cpu().numpy().view(gpsacq.PEAK_DTYPE), want)
        assert np.array_equal(half.cpu().numpy().view(gpsacq.PEAK_DTYPE), want[:n // 2])


def test_noncoherent_creep_compensation():
    """Weak satellite at a large carrier offset (59.99 kHz at fs 2.8 MHz: the code creeps 1.6 samples
    per accumulated block): with gpsacq_set_creep_compensation the five blocks' peaks line up again.
    The compensated sums equal the oracle's restatement (integer re-alignment per block and bin)."""
    import gpsacq
    from oracle_lib import Oracle
    fc, fs = 0.62e6, 2.8e6
    with gpsacq.Engine(fc, fs, 100000.0) as eng:
        stride = eng.aligned_stride()
        assert stride == 5250
        fd = 857 * fs / 40000
        bits = eng.generate(5 * stride + 5120, [(9, 0.075, fd, 777.0, 0.3)], noise_sigma=1.0, seed=99)
        eng.set_doppler_window(800, 120)
        eng.set_noncoherent(5, 1)
        tasks = [(0, 8), (0, 30)]
        c_off, p_off = eng.search(bits, tasks=tasks, stride=stride)
        eng.set_creep_compensation(True)
        c_on, p_on = eng.search(bits, tasks=tasks, stride=stride)
        assert int(p_on["lo_shift"][0]) == 857 and abs(int(p_on["ca_shift"][0]) - 777) <= 1
        assert p_on["snr"][0] > 1.25 * p_off["snr"][0], (p_on["snr"], p_off["snr"])
        assert p_on["snr"][0] > 2.0 * p_on["snr"][1]  # absent PRN 31
        assert np.allclose(c_on["tot_pwr"], c_off["tot_pwr"], rtol=1e-5)  # a permutation of the same powers
        orc = Oracle(fc, fs, 100000.0)
        want = orc.search_noncoherent(bits, stride, 0, 8, 5, 1, first_bin=840, n_bins=30, creep=True)
        got = c_on[0][40:70]
        np.testing.assert_allclose(got["max_pwr"], want["max_pwr"], rtol=2e-5)
        np.testing.assert_allclose(got["tot_pwr"], want["tot_pwr"], rtol=2e-5)
        assert (got["max_i"] != want["max_i"]).sum() <= 1
        # coherent searches ignore the switch
        eng.set_noncoherent(1)
        a, _ = eng.search(bits, tasks=tasks, stride=stride)
        eng.set_creep_compensation(False)
        b, _ = eng.search(bits, tasks=tasks, stride=stride)
        assert np.array_equal(a, b)


def test_noncoherent_block_alignment_any_stride(golden_dir):
    """gpsacq_set_block_alignment (SURVEY.md section 8d configs[3], "per-block lag re-alignment"): the file's own contiguous
    5120-byte blocks are 40960 samples apart -- 40960 mod 5456 = 2768 samples of code phase -- so a plain non-coherent sum smears the
    peak over five lags; with alignment on the weak PRN 12 stands out as with the aligned layout, the sums equal the oracle's
    restatement, and the code phase refers to block 0.  Also together with creep re-alignment, at a rate with two column passes, and
    on an 8-bit IQ capture (81920-byte blocks)."""
    import gpsacq
    from oracle_lib import Oracle
    fc, fs = 4.092e6, 5.456e6
    orc = Oracle(fc, fs, 5000.0)
    with gpsacq.Engine(fc, fs, 5000.0) as eng:
        # the same signal parameters as the weak-signal fixture, generated contiguously
        bits = eng.generate(6 * 5120, [(12, 0.06, 4 * fs / 40000, 1000.0, 0.1), (3, 0.2, -11 * fs / 40000, 3333.0, 0.7)], noise_sigma=1.0, seed=5)
        tasks = [(0, 11), (0, 2), (0, 20)]
        eng.set_noncoherent(5, 1)
        c_plain, p_plain = eng.search(bits, tasks=tasks)
        eng.set_block_alignment(True)
        c_al, p_al = eng.search(bits, tasks=tasks)
        for t, (b, sv) in enumerate(tasks):
            want = orc.search_noncoherent(bits, 5120, b, sv, 5, 1, align=True)
            np.testing.assert_allclose(c_al["max_pwr"][t], want["max_pwr"], rtol=2e-5)
            np.testing.assert_allclose(c_al["tot_pwr"][t], want["tot_pwr"], rtol=2e-5)
            assert (c_al["max_i"][t] != want["max_i"]).sum() <= 1
        assert np.allclose(c_al["tot_pwr"], c_plain["tot_pwr"], rtol=1e-5)  # a permutation of the same powers
        assert int(p_al["lo_shift"][0]) == 4 and abs(int(p_al["ca_shift"][0]) - 1000) <= 1
        assert int(p_al["lo_shift"][1]) == -11 and int(p_al["ca_shift"][1]) == 3333
        assert p_al["snr"][0] > 2.0 * p_al["snr"][2] and p_al["snr"][0] > 1.5 * p_plain["snr"][0], (p_al["snr"], p_plain["snr"])
        # with the creep re-alignment on top, every second block
        eng.set_noncoherent(3, 2)
        eng.set_creep_compensation(True)
        c2, _ = eng.search(bits, tasks=[(0, 2)])
        want = orc.search_noncoherent(bits, 5120, 0, 2, 3, 2, creep=True, align=True)
        np.testing.assert_allclose(c2["max_pwr"][0], want["max_pwr"], rtol=2e-5)
        assert (c2["max_i"][0] != want["max_i"]).sum() <= 1
        # off again: the plain sums, bit for bit
        eng.set_creep_compensation(False)
        eng.set_block_alignment(False)
        eng.set_noncoherent(5, 1)
        c3, _ = eng.search(bits, tasks=tasks)
        assert np.array_equal(c3, c_plain)
    # 12 MHz: 12000 lags in two column passes (per-lag sums in device memory), blocks 40960 samples = 3.41 periods apart
    fc, fs = 3.0e6, 12.0e6
    orc = Oracle(fc, fs, 3000.0)
    with gpsacq.Engine(fc, fs, 3000.0) as eng:
        bits = eng.generate(4 * 5120, [(7, 0.08, 2 * fs / 40000, 11990.0, 0.2)], noise_sigma=1.0, seed=9)
        eng.set_noncoherent(4, 1)
        eng.set_block_alignment(True)
        c, p = eng.search(bits, tasks=[(0, 6), (0, 19)])
        want = orc.search_noncoherent(bits, 5120, 0, 6, 4, 1, align=True)
        np.testing.assert_allclose(c["max_pwr"][0], want["max_pwr"], rtol=2e-5)
        np.testing.assert_allclose(c["tot_pwr"][0], want["tot_pwr"], rtol=2e-5)
        assert (c["max_i"][0] != want["max_i"]).sum() <= 1
        assert int(p["lo_shift"][0]) == 2 and abs(int(p["ca_shift"][0]) - 11990) <= 1 and p["snr"][0] > 2.0 * p["snr"][1]
    # a rate whose code period is not a whole number of samples is refused
    with gpsacq.Engine(1.0e6, 4.0005e6, 3000.0) as eng:
        eng.set_noncoherent(2, 1)
        eng.set_block_alignment(True)
        with pytest.raises(gpsacq.GpsAcqError):
            eng.search(np.zeros(3 * 5120, np.uint8))


def test_noncoherent_creep_compensation_beyond_10000_lags():
    """fs = 16.368 MHz: 16 368 lags are searched in two passes of 40 columns, and a lag's re-aligned destination can lie in the
    other pass's window -- the per-lag sums then live in device memory (k_corr's corr_dump_power + k_scan_power).  A satellite at
    a large carrier offset lines up over five blocks; compensated sums against the oracle's restatement, including the bins whose
    shift wraps around the code period; tot_pwr is a permutation of the uncompensated search's."""
    import gpsacq
    from oracle_lib import Oracle
    fc, fs = 4.092e6, 16.368e6
    with gpsacq.Engine(fc, fs, 100000.0) as eng:
        stride = eng.aligned_stride()
        assert stride == 6138 and eng.num_lags == 16368  # three code periods of 2046 bytes
        bin_hz = fs / 40000
        fd = 215 * bin_hz  # 87.98 kHz: the code creeps 2.7 samples per accumulated block of 49 104 samples
        bits = eng.generate(5 * stride + 5120, [(9, 0.075, fd, 9990.3, 0.3)], noise_sigma=1.0, seed=7)
        eng.set_doppler_window(190, 50)
        eng.set_noncoherent(5, 1)
        tasks = [(0, 8), (0, 30)]
        c_off, p_off = eng.search(bits, tasks=tasks, stride=stride)
        eng.set_creep_compensation(True)
        c_on, p_on = eng.search(bits, tasks=tasks, stride=stride)
        assert eng.last_timing()["correlate_launches"] == 2
        assert int(p_on["lo_shift"][0]) == 215 and abs(int(p_on["ca_shift"][0]) - 9990) <= 1  # next to the 10000-lag pass boundary
        # (a chip is 16 samples wide here: the 11 samples crept over the five blocks cost less than at 2.8 MHz)
        assert p_on["snr"][0] > 1.1 * p_off["snr"][0], (p_on["snr"], p_off["snr"])
        assert p_on["snr"][0] > 2.0 * p_on["snr"][1]
        assert np.allclose(c_on["tot_pwr"], c_off["tot_pwr"], rtol=1e-5)
        orc = Oracle(fc, fs, 100000.0)
        want = orc.search_noncoherent(bits, stride, 0, 8, 5, 1, first_bin=205, n_bins=20, creep=True)
        got = c_on[0][15:35]
        np.testing.assert_allclose(got["max_pwr"], want["max_pwr"], rtol=2e-5)
        np.testing.assert_allclose(got["tot_pwr"], want["tot_pwr"], rtol=2e-5)
        assert (got["max_i"] != want["max_i"]).sum() <= 1
        # negative offsets: the shift has the other sign and wraps below lag 0
        eng.set_doppler_window(-230, 12)
        c_neg, _ = eng.search(bits, tasks=[(0, 30)], stride=stride)
        want = orc.search_noncoherent(bits, stride, 0, 30, 5, 1, first_bin=-230, n_bins=12, creep=True)
        np.testing.assert_allclose(c_neg["max_pwr"][0], want["max_pwr"], rtol=2e-5)
        np.testing.assert_allclose(c_neg["tot_pwr"][0], want["tot_pwr"], rtol=2e-5)
        assert (c_neg["max_i"][0] != want["max_i"]).sum() <= 1


@pytest.mark.parametrize("fc,fs,cols", [(2.046e6, 8.184e6, 33), (2.6e6, 10e6, 40)])
def test_noncoherent_wide_instances(golden_dir, fc, fs, cols):
    """The 33- and 40-column non-coherent instances (the only users of the 40 KB LDS slot map in k_corr: their per-lag
    power array leaves no room for the 45 KB one) against the oracle's restatement."""
    import gpsacq
    from oracle_lib import Oracle
    orc = Oracle(fc, fs, 5000.0)
    with gpsacq.Engine(fc, fs, 5000.0) as eng:
        assert eng.acc_columns == cols
        stride = eng.aligned_stride()
        assert stride % (eng.num_lags // 8) == 0 and stride >= 5120
        buf = eng.generate(4 * stride + 5120, [(8, 0.4, 1200.0, 777.0, 0.1), (19, 0.3, -2600.0, 4321.0, 0.6)], noise_sigma=1.0, seed=5).tobytes()
        eng.set_noncoherent(3, 1)
        tasks = [(0, 7), (1, 18), (0, 2)]
        cells, peaks = eng.search(buf, tasks=tasks, stride=stride)
        for t, (b, sv) in enumerate(tasks):
            want = orc.search_noncoherent(buf, stride, b, sv, 3, 1)
            np.testing.assert_allclose(cells["max_pwr"][t], want["max_pwr"], rtol=2e-5)
            np.testing.assert_allclose(cells["tot_pwr"][t], want["tot_pwr"], rtol=2e-5)
            assert (cells["max_i"][t] != want["max_i"]).sum() <= 1
        assert peaks["snr"][0] > 2 * peaks["snr"][2] and peaks["snr"][1] > 2 * peaks["snr"][2]


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_random_grid_configurations(seed):
    """Seeded random (IF, sampling rate, Doppler range, Doppler step) combinations on random capture bits: ten random grid
    points of two random (block, PRN) tasks against the oracle's restatement of the grid (Oracle.search_grid), and the peak
    against the scan over all the engine's own cells."""
    import gpsacq
    from oracle_lib import Oracle
    rng = np.random.default_rng(1000 + seed)
    fs = float(rng.uniform(2.0e6, 10.0e6))
    fc = float(rng.uniform(0.0, 0.45 * fs))
    max_fo = float(rng.uniform(500.0, 20000.0))
    bin_hz = fs / 40000.0
    step = [0.0, bin_hz / 2, bin_hz / 3, 2.5 * bin_hz, bin_hz / 5, 4.2 * bin_hz][seed - 1]
    bits = rng.integers(0, 256, size=3 * 5120, dtype=np.uint8).tobytes()
    orc = Oracle(fc, fs, max_fo)
    with gpsacq.Engine(fc, fs, max_fo) as eng:
        eng.set_doppler_step(step)
        sub, stride, kmax = eng.doppler_sub, eng.doppler_stride, eng.kmax
        assert eng.num_doppler == 2 * kmax + 1 and kmax == int(max_fo / eng.doppler_step_hz)
        tasks = [(int(rng.integers(0, 3)), int(rng.integers(0, 32))) for _ in range(2)]
        cells, peaks = eng.search(bits, tasks=tasks)
        for t, (b, sv) in enumerate(tasks):
            pts = sorted(set([-kmax, kmax, 0] + [int(v) for v in rng.integers(-kmax, kmax + 1, 7)]))
            oc, ks = orc.search_grid(bits[b * 5120:(b + 1) * 5120], sv, sub=sub, dstride=stride, points=pts)
            got = cells[t][np.array(ks) + kmax]
            np.testing.assert_allclose(got["max_pwr"], oc["max_pwr"], rtol=2e-5)
            np.testing.assert_allclose(got["tot_pwr"], oc["tot_pwr"], rtol=2e-5)
            assert np.array_equal(got["max_i"], oc["max_i"])
            k = int(np.argmax(cells[t]["snr"]))  # first maximum == strict '>' scan over ascending frequency
            assert peaks["lo_shift"][t] == k - kmax and peaks["ca_shift"][t] == cells[t]["max_i"][k]
