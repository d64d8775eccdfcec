"""Round 6 on the GPU box: every world size searches THE SAME capture and says so in a digest of the merged keys; rank 0's share is
checked against the oracle at any N; the C ABI's own multi-GPU entry point meets the same capture; the same cells through rocFFT;
all 32 PRNs of the device chip table against the reference-held IS-GPS-200G Table 3-I."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(**kw):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GPSACQ_DIST_BACKEND"):
        e.pop(k, None)
    e.update(kw)
    return e


def _line(r):
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout[-2000:] + r.stderr[-3000:]
    return json.loads(lines[0])


def _bench(*args, **env):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=1200, env=_env(**env))
    assert r.returncode == 0, r.stderr[-3000:]
    return _line(r)


def _torchrun(n, *args):
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(n), *args],
                       capture_output=True, text=True, timeout=1500, env=_env(GPSACQ_DIST_BACKEND="gloo"))
    assert r.returncode == 0, r.stderr[-3000:]
    return _line(r)


def test_generator_ranges_are_windows_of_one_stream():
    """gpsacq_generate_range: the bytes that start at sample first_sample are the bytes gpsacq_generate writes there -- noise and
    signals are functions of the absolute sample index -- so any rank can make exactly its own blocks of the one capture."""
    import gpsacq
    sats = [(3, 0.2, 1234.5, 100.25, 0.3), (17, 0.151, -3100.0, 4000.0, 0.9)]
    with gpsacq.Engine(4.092e6, 5.456e6, 5000.0) as eng:
        whole = eng.generate(40 * 5120, sats, noise_sigma=1.0, seed=77)
        for blk, n in ((0, 3), (7, 5), (32, 8), (39, 1)):
            part = eng.generate(n * 5120, sats, noise_sigma=1.0, seed=77, first_sample=blk * 5120 * 8)
            assert np.array_equal(part, whole[blk * 5120:(blk + n) * 5120]), blk
        odd = eng.generate(1000, sats, noise_sigma=1.0, seed=77, first_sample=8 * 12345)
        assert np.array_equal(odd, whole[12345:13345])
        assert not np.array_equal(eng.generate(5120, sats, seed=78), whole[:5120])
        with pytest.raises(gpsacq.GpsAcqError):
            eng.generate(16, sats, first_sample=3)  # not a byte boundary


def test_device_chip_table_against_is_gps_200_table_3_I(golden_dir):
    """The chips the device generators read (c_chips, uploaded by gpsacq_create from the product's CaCode) for all 32 PRNs: the
    script's transmit signal (k_siggen_tx, gps_sig_gen.m:21-30) reads 50 x chip at every chip centre (sample 24 + 8 j), so its first
    ten centres spell the octal column of the reference-held IS-GPS-200G Table 3-I and its first period is G1 xor G2 delayed by the
    table's chip delay (c/cacode.h:9-35, c/search_offline.cpp:20-53)."""
    import gpsacq
    from test_oracle import _table_3_I, _g1_g2
    first10, delays = _table_3_I(golden_dir)
    g1, g2 = _g1_g2()
    with gpsacq.Engine(0.0, 8.184e6, 5000.0) as eng:
        for prn in range(1, 33):
            iq = eng.generate_sig_tx(prn, [1], n_repeat=1, first_sample=0, n_samples=24 + 8 * 1023)
            centres = iq[0::2][24::8][:1023].astype(np.int32)
            assert set(np.unique(centres)) <= {-50, 50}, prn
            chips = (centres < 0).astype(np.uint8)  # chip 1 -> -1.0 (Bipolar, c/search_offline.cpp:68-70)
            assert list(chips[:10]) == first10[prn - 1], prn
            assert np.array_equal(chips, g1 ^ np.roll(g2, delays[prn - 1])), prn


SMALL = ("--blocks-total", "1280", "--steps", "3", "--warmup", "1", "--weak-blocks", "0", "--no-e2e", "--no-live-traffic", "--soak-seconds", "0",
         "--no-library-baseline")


def test_keys_digest_is_identical_at_1_2_and_8_ranks_and_rank0_is_checked_against_the_oracle():
    """north_star: "identical acquisition results ... at 1/2/4/8".  A 40-run capture searched by 1, 2 and 8 ranks (gloo collectives, all
    ranks on the one GPU; the driver's torchrun line): `keys_digest` -- sha256 of the 32 merged per-PRN keys -- and `detected` are the
    same, bit for bit; at every N rank 0's share carries a green parity verdict against the oracle; at N > 1 the C ABI's own
    gpsacq_multi_search_blocks (one process, one engine per rank's device) reproduces the keys over the same capture."""
    j1 = _bench(*SMALL)
    assert j1["n_gpus"] == 1 and j1["keys_digest_comparable_across_n"] is True and len(j1["keys_digest"]) == 16
    assert j1["cpu_baseline"]["parity_ok"] is True and j1["cpu_baseline"]["parity_vs_gpu"]["first_block_of_this_rank"] == 0
    assert j1["cpu_baseline"]["parity_blocks"] == 1280 and j1["cpu_baseline"]["parity_cells"] == 1280 * 73 and j1["cpu_baseline"]["parity_whole_share"] is True
    assert set(j1["detected_prns"]) >= set(j1["injected_prns_all_ranks"]) and len(j1["injected_prns_all_ranks"]) == 8
    for n in (2, 8):
        j = _torchrun(n, *SMALL)
        assert j["n_gpus"] == n and j["rccl_ranks_seen"] == n and sum(j["blocks_per_rank"]) == 1280 and j["devices_per_rank"] == [0] * n
        assert j["keys_digest"] == j1["keys_digest"], (n, j["detected"], j1["detected"])
        assert j["detected"] == j1["detected"] and j["injected_prns_all_ranks"] == j1["injected_prns_all_ranks"]
        assert "cpu_baseline" not in j  # an N = 1 figure
        par = j["parity_vs_gpu"]
        # rank 0's WHOLE share: every block's peak and every cell of the last timed step against the oracle
        assert j["parity_ok"] is True and par["ok"] is True and par["whole_share"] is True and j["parity_whole_share"] is True, par
        assert par["blocks"] == j["blocks_per_rank"][0] and par["cells"] == par["blocks"] * 73 and par["first_block_of_this_rank"] == 0, par
        assert par["snr_max_rel"] <= 1e-4 and par["pwr_max_rel"] <= 2e-5 and not par["cell_lag_mismatches"] and not par["cell_power_mismatches"]
        assert "error" not in j["parity_oracle_run"], j["parity_oracle_run"]
        im = j["extras"]["inproc_multi"]
        assert "error" not in im, im
        assert im["keys_equal_digest"] is True and im["keys_digest"] == j1["keys_digest"] and im["devices"] == [0] * n and im["runs"] == 40


def test_rank_without_work_contributes_zero_keys_every_step():
    """More ranks than runs: the rank that owns no run must hand zeros to the all-reduce in EVERY step (its key buffer holds the previous
    merged keys otherwise: ADVICE r5) -- the digest of a 2-run capture on 3 ranks equals the 1-rank digest."""
    args = ("--blocks-total", "64", "--steps", "4", "--warmup", "1", "--bare")
    j1 = _bench(*args)
    j3 = _torchrun(3, *args)
    assert j3["blocks_per_rank"] == [32, 32, 0]
    assert j3["keys_digest"] == j1["keys_digest"] and j3["detected"] == j1["detected"]


def test_line_carries_the_rocfft_baseline_and_the_colimiters():
    """extras.gpu_library_baseline: the same cells through rocFFT (torch.fft) on the same GPU -- a figure beside `value`, and its
    detections agree with k_corr's (same code phase, same Doppler bin); roofline carries the three co-limiters (VALU / LDS / L2 -> L1
    at 128-byte lines) and the energy per cell."""
    j = _bench("--blocks-total", "640", "--steps", "4", "--warmup", "1", "--weak-blocks", "0", "--no-e2e", "--no-live-traffic", "--soak-seconds", "0",
               "--no-cpu-baseline")
    lb = j["extras"]["gpu_library_baseline"]
    assert "error" not in lb, lb
    assert lb["cells"] == 32 * 73 and lb["cells_per_s"] > 0 and lb["cells_per_s_ifft_only"] > lb["cells_per_s"]
    ag = lb["agrees_with_k_corr"]
    assert ag["detections"] >= 1 and ag["ca_equal_on_detections"] == ag["detections"] and ag["lo_equal_on_detections"] == ag["detections"]
    assert ag["snr_max_rel"] < 1e-3
    assert j["roofline"]["kernel_cells_per_s"] > lb["cells_per_s"]  # the fused kernel beats product + library transform + scan
    rf = j["roofline"]
    tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    if tj.get("onchip_counters", {}).get("l2_to_l1_bytes_per_cell"):
        assert 600e3 < rf["l2_to_l1_bytes_per_cell"] < 1000e3  # both spectra + twiddles + tables through L1 at 128-byte lines: ~0.8 MB per cell
        assert 0.2 < rf["l2_frac"] < 0.9 and 0.2 < rf["lds_frac"] < 0.9 and 0.3 < rf["valu_busy_frac"] < 1.0
    assert rf["energy_uj_per_cell"] is None or 10 < rf["energy_uj_per_cell"] < 500
    assert rf["clock_sampling"]["closing_stamp_kernel_ms"] < 0.5
