"""Round 5 on the GPU box: the bench line verifies itself against the oracle at the configuration it measures and carries a
box-independent roofline; the multi-GPU merge keys are made by the library."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_bench(*args):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GPSACQ_DIST_BACKEND"):
        e.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=900, env=e)


def _line(r):
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:] + r.stderr[-2000:]
    return json.loads(lines[0])


def test_default_line_carries_its_own_parity_verdict_and_clock():
    """cpu_baseline.parity_vs_gpu: the peaks of the last timed step against the oracle's float build on EVERY block of the step (the
    all-cores pass of the oracle; c/search_offline.cpp:190-198: same code phase, same Doppler bin, SNR to 1e-4, ties proven in double),
    and every cell of those blocks (powers to 1e-5 of the float build or 2e-5 of liboracle_f64, lags equal or proven ties); roofline.sclk_mhz / cycles_per_cell_per_cu / frac_at_clock from readings
    taken DURING the timed steps; roofline.pk_fma_stream_TF from the micro-benchmark run in the untimed part."""
    r = _run_bench("--steps", "8", "--warmup", "2", "--blocks-total", "640", "--weak-blocks", "0", "--no-e2e", "--no-live-traffic",
                   "--soak-seconds", "0.5", "--pk-fma-seconds", "1.0")
    assert r.returncode == 0, r.stderr[-3000:]
    j = _line(r)
    cb = j["cpu_baseline"]
    par = cb["parity_vs_gpu"]
    assert par["ok"] is True and cb["parity_ok"] is True, par
    assert par["blocks"] == 640 and par["whole_share"] is True and par["ca_equal"] + par["proven_ties"] >= par["blocks"] and not par["unproven_mismatches"]
    assert par["lo_equal"] + par["proven_ties"] >= par["blocks"]
    assert par["snr_max_rel"] <= 1e-4 and par["pwr_max_rel"] <= 2e-5 and par["cells"] == 640 * 73 and not par["cell_lag_mismatches"]
    assert par["non_finite_or_non_positive_cell_powers"] == 0 and not par["cell_power_mismatches"] and not par["too_many_cells_off_the_screen"]
    ac = j["cpu_baseline_all_cores"]
    assert "error" not in ac and ac["blocks"] == 640 and ac["equals_the_one_thread_run"] is True and ac["value"] > 0, ac
    assert cb["parity_blocks"] == par["blocks"] and cb["parity_snr_max_rel"] == par["snr_max_rel"] and cb["parity_pwr_max_rel"] == par["pwr_max_rel"]
    rf = j["roofline"]
    cs = rf["clock_sampling"]
    assert cs["samples"] >= 1 and cs["sclk_mhz_sysfs"] is not None, cs
    # the roofline's clock is the GPU's own cycle count over the timed steps (gpsacq_cycle_stamp_device), not the lagging sysfs average
    assert cs["sclk_mhz_cycle_counter"] == rf["sclk_mhz"] and 500 <= rf["sclk_mhz"] <= 2500, cs
    per = cs["sclk_mhz_per_xcd"]  # every XCD has its own counter and clock: all eight stamped, within 15 % of each other
    assert len(per) == 8 and all(p is not None for p in per) and max(per) / min(per) < 1.15, per
    assert rf["sclk_mhz_xcd_min"] == min(per) and rf["sclk_mhz_xcd_max"] == max(per) and min(per) <= rf["sclk_mhz"] <= max(per)
    assert rf["power_w"] is None or rf["power_w"] > 50
    want = rf["kernel_ms"] * 1e-3 * rf["sclk_mhz"] * 1e6 * rf["compute_units"] / rf["cells_per_launch"]
    assert abs(rf["cycles_per_cell_per_cu"] / want - 1) < 1e-9 and 10000 < rf["cycles_per_cell_per_cu"] < 100000
    assert abs(rf["frac_at_clock"] - rf["achieved"] / (157.3 * rf["sclk_mhz"] / 2400.0)) < 1e-9
    assert rf["frac"] <= rf["frac_at_clock"] * 1.03 and rf["frac_at_clock"] < 1  # (a 20 ms timed region may sit at the 2.4 GHz boost clock)
    assert 60 < rf["pk_fma_stream_TF"] < 158 and 0 < rf["frac_of_pk_fma_stream"] < 1, rf["pk_fma_stream"]
    ex = j["extras"]
    assert ex["one_rank_process_group"] and ex["pk_fma_stream"] and ex["cpu_baseline"] and not ex["live_traffic"] and not ex["e2e_cli"]
    oc = j["one_rank_collective"]
    assert oc["ms_per_step_without_process_group"] > 0 and oc["ms_per_step_with"] == j["ms_per_step"]


def test_a_wrong_gpu_result_fails_the_bench_run():
    """--parity-selftest moves one GPU peak by one lag before the comparison: the line still comes out, says so, and the exit code is 3."""
    r = _run_bench("--steps", "2", "--warmup", "1", "--blocks-total", "640", "--weak-blocks", "0", "--no-e2e", "--no-live-traffic",
                   "--soak-seconds", "0", "--parity-selftest")
    assert r.returncode == 3, (r.returncode, r.stderr[-1500:])
    j = _line(r)
    par = j["cpu_baseline"]["parity_vs_gpu"]
    assert par["ok"] is False and par["unproven_mismatches"] == [7] and j["cpu_baseline"]["parity_ok"] is False
    assert "DISAGREE" in r.stderr


def test_peak_keys_device_equals_the_torch_packing(golden_dir):
    """gpsacq_peak_keys_device (one launch on the engine's stream) = gpsacq.dist.pack_keys / per_prn_best bit for bit: per-PRN best
    keys of the reference schedule, per-task keys, and 32 zero keys for a rank without work."""
    import torch
    import gpsacq
    from gpsacq import dist as D
    buf = np.frombuffer(open(os.path.join(golden_dir, "synth_nott_fs5456.bin"), "rb").read(), dtype=np.uint8)
    nblk = buf.size // 5120
    nblk -= nblk % 32
    dev = torch.device("cuda", 0)
    with gpsacq.Engine(4.092e6, 5.456e6, 5000.0) as eng:
        d_bits = torch.from_numpy(buf[:nblk * 5120].copy()).to(dev)
        d_peaks = torch.zeros((nblk, 4), dtype=torch.int32, device=dev)
        k32 = torch.full((32,), -1, dtype=torch.int64, device=dev)
        kall = torch.full((nblk,), -1, dtype=torch.int64, device=dev)
        torch.cuda.synchronize()  # the fills run on torch's stream, the search on the engine's own (non-blocking) stream: order them
        eng.search_device(d_bits.data_ptr(), nblk, d_peaks.data_ptr(), sync=False)
        eng.peak_keys_device(d_peaks.data_ptr(), nblk, k32.data_ptr(), per_prn=True)
        eng.peak_keys_device(d_peaks.data_ptr(), nblk, kall.data_ptr(), per_prn=False, sync=True)
        want_all = D.pack_keys(d_peaks, eng.kmax)
        assert torch.equal(kall, want_all) and torch.equal(k32, D.per_prn_best(want_all))
        snr, lo, ca = D.unpack_keys(k32.cpu(), eng.kmax)
        assert [int(p) + 1 for p in torch.nonzero(snr >= 25).flatten()] == [1, 21, 29, 30, 31]
        # a partial last run: tasks 0 .. 39 -> PRNs 0..7 see two candidates, the others one
        eng.peak_keys_device(d_peaks.data_ptr(), 40, k32.data_ptr(), per_prn=True, sync=True)
        w = torch.zeros(32, dtype=torch.int64, device=dev)
        w[:] = want_all[:32]
        w[:8] = torch.maximum(w[:8], want_all[32:40])
        assert torch.equal(k32, w)
        eng.peak_keys_device(0, 0, k32.data_ptr(), per_prn=True, sync=True)  # a rank without work
        assert int(k32.abs().sum()) == 0


def test_headline_size_properties_on_the_device_path():
    """BASELINE configs[1] at the size the metric is quoted on -- 340 runs = 10 880 blocks x 73 bins = 794 240 cells in ONE device
    search, where the oracle is out of reach (4 minutes of CPU): size-independent properties instead.  The batch equals its two
    halves bit for bit (170 runs each: the reference schedule block -> PRN block % 32 holds in both), a rerun is bit-identical,
    the library's per-PRN keys equal the torch restatement, every injected satellite is found in EVERY run at the Doppler bin it
    was generated with and at the code phase law of SearchTask's schedule (ca_shift advances by 40960 samples per block,
    c/search_offline.cpp:239-246), and the PRNs that were not injected stay at the noise level (threshold of :248) in every run."""
    import torch
    import gpsacq
    from gpsacq import dist as D
    fs, fc, S = 5.456e6, 4.092e6, 5456
    rs = np.random.default_rng(1000)
    prns = sorted(rs.choice(np.arange(1, 33), size=8, replace=False).tolist())
    sats = [(p, 0.151, float(rs.uniform(-4500, 4500)), float(rs.uniform(0, S)), float(rs.random())) for p in prns]
    nblk, dev = 10880, torch.device("cuda", 0)
    with gpsacq.Engine(fc, fs, 5000.0) as eng:
        nbytes = nblk * 5120
        d_bits = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        eng.generate_device(d_bits.data_ptr(), nbytes, sats, noise_sigma=1.0, seed=1000)
        pk = [torch.zeros((nblk, 4), dtype=torch.int32, device=dev) for _ in range(3)]
        eng.search_device(d_bits.data_ptr(), nblk, pk[0].data_ptr(), sync=True)
        assert eng.last_timing()["cells"] == nblk * 73
        eng.search_device(d_bits.data_ptr(), nblk, pk[1].data_ptr(), sync=True)
        assert torch.equal(pk[0], pk[1])  # deterministic
        half = nblk // 2  # 5440 blocks = 170 whole runs
        eng.search_device(d_bits.data_ptr(), half, pk[2].data_ptr(), sync=True)
        eng.search_device(d_bits[half * 5120:].data_ptr(), half, pk[2][half:].data_ptr(), sync=True)
        assert torch.equal(pk[0], pk[2])  # a batch equals its halves
        keys = torch.zeros(32, dtype=torch.int64, device=dev)
        eng.peak_keys_device(pk[0].data_ptr(), nblk, keys.data_ptr(), per_prn=True, sync=True)
        assert torch.equal(keys, D.per_prn_best(D.pack_keys(pk[0], eng.kmax)))
        peaks = pk[0].cpu().numpy().view(gpsacq.PEAK_DTYPE).reshape(nblk // 32, 32)
    for prn, amp, dop, ca, ph in sats:
        col = peaks[:, prn - 1]
        assert col["snr"].min() > 30, (prn, float(col["snr"].min()))
        assert np.abs(col["lo_shift"] - round(dop * 40000 / fs)).max() <= 1
        blocks = 32 * np.arange(nblk // 32) + (prn - 1)
        expect = (ca + 40960.0 * blocks * (1 + dop / 1575.42e6)) % S
        d = np.abs(col["ca_shift"] - expect)
        assert np.minimum(d, S - d).max() <= 1.5, (prn, float(np.minimum(d, S - d).max()))
    absent = [sv for sv in range(32) if sv + 1 not in prns]
    # 8160 noise-only (run, PRN) peaks, each the largest of 73 x 5456 exponentially distributed ratios: P(one of them >= 25) is a few
    # per cent for a random seed -- the seed is fixed, but the bound leaves that much room instead of asserting a coin flip
    noise = peaks[:, absent]["snr"]
    assert (noise >= 25).sum() <= 1 and noise.max() < 30, float(noise.max())


def test_parity_verdict_on_the_references_own_capture(golden_dir):
    """`bench.py --config 2 --capture gps_sig_tmp.bin` with the CPU baseline on: the verdict covers every one of the 384 blocks of the
    reference-held file (12 runs x 32 PRN x 49 bins at fs 8.184 MHz, 8184 lags; the k_corr<33> instance) -- same code phase and Doppler
    bin as the oracle in all of them, every one of their 18 816 cells within tolerance -- and PRN 8 is the file's satellite (README.md:45,57)."""
    r = _run_bench("--config", "2", "--capture", os.path.join(golden_dir, "gps_sig_tmp.bin"), "--steps", "2", "--warmup", "1", "--no-e2e",
                   "--no-live-traffic", "--soak-seconds", "0", "--weak-blocks", "0")
    assert r.returncode == 0, r.stderr[-3000:]
    j = _line(r)
    par = j["cpu_baseline"]["parity_vs_gpu"]
    assert par["ok"] is True and par["blocks"] == 384 and par["ca_equal"] + par["proven_ties"] == 384 and par["cells"] == 384 * 49 and par["whole_share"] is True
    assert par["snr_max_rel"] <= 1e-4 and par["pwr_max_rel"] <= 2e-5
    best = {d["prn"]: d for d in j["detected"]}
    assert 8 in best and best[8]["lo_shift"] == 0 and best[8]["snr"] > 500
    assert j["roofline"]["kernel"] == "k_corr<33>"
