"""bench.py's measurement helpers that need no GPU: the sysfs clock / power reader behind roofline.sclk_mhz and the sampler's
statistics (CPU suite; the GPU box runs them for real in tests/test_gpu_round5.py)."""
import importlib.util
import os
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_sysfs_clock_and_power_reader(tmp_path):
    b = _bench()
    card = tmp_path / "card0" / "device"
    (card / "hwmon" / "hwmon3").mkdir(parents=True)
    (card / "pp_dpm_sclk").write_text("0: 132Mhz \n1: 2214Mhz *\n")  # the format of an MI355X box (gpurun_out/r05a/sysfs.txt)
    (card / "hwmon" / "hwmon3" / "power1_average").write_text("1363000000\n")
    assert b.read_clock_power(str(card)) == {"sclk_mhz": 2214, "power_w": 1363.0}
    (card / "hwmon" / "hwmon3" / "power1_average").unlink()
    (card / "hwmon" / "hwmon3" / "power1_input").write_text("901500000\n")
    assert b.read_clock_power(str(card)) == {"sclk_mhz": 2214, "power_w": 901.5}
    (card / "pp_dpm_sclk").write_text("garbage\n")
    assert b.read_clock_power(str(card))["sclk_mhz"] is None
    assert b.read_clock_power(str(tmp_path / "nothing")) == {"sclk_mhz": None, "power_w": None}


def test_clock_sampler_statistics(tmp_path):
    """Medians over the readings taken between start() and stop(); the first one 10 ms in, so a leg of a few steps gets one."""
    b = _bench()
    card = tmp_path / "device"
    (card / "hwmon" / "hwmon0").mkdir(parents=True)
    (card / "pp_dpm_sclk").write_text("0: 132Mhz\n1: 2100Mhz *\n")
    (card / "hwmon" / "hwmon0" / "power1_average").write_text("1000000000\n")
    s = b.ClockSampler.__new__(b.ClockSampler)
    s.card, s.dev_index, s.period = str(card), 0, 0.02
    s.start()
    time.sleep(0.05)
    (card / "pp_dpm_sclk").write_text("0: 132Mhz\n1: 2200Mhz *\n")
    time.sleep(0.12)
    s.stop()
    st = s.stats()
    assert st["samples"] >= 4 and st["sclk_mhz_min_max"] == [2100, 2200] and 2100 <= st["sclk_mhz"] <= 2200 and st["power_w"] == 1000.0
    assert "sysfs" in st["source"]
    s.start()  # a leg shorter than the period still gets its reading
    time.sleep(0.03)
    s.stop()
    assert s.stats()["samples"] >= 1


def test_flops_and_peak_constants():
    b = _bench()
    assert abs(b.flops_per_cell(5456) - (6 * 40000 + 5 * 40000 * 15.287712379549449 + 5 * 5456)) < 1.0
    # 157.3 TFLOP/s = 256 CUs x 128 FMA lanes x 2 flop x 2.4 GHz: the clock frac_at_clock normalises to
    assert abs(256 * 128 * 2 * b.FP32_PEAK_CLOCK_MHZ * 1e6 / 1e12 - b.FP32_VALU_PEAK_TF) < 0.05


def test_compare_peaks_verdict_logic(monkeypatch):
    """The host half of cpu_baseline.parity_vs_gpu on the oracle alone (no GPU): the double-precision oracle's peaks stand in for the
    GPU's.  Equal results pass; a peak moved to another lag is an UNPROVEN mismatch (the bench run then exits 3); the same move is
    accepted as a tie only when the two candidates' double-precision SNRs agree to the tie tolerance (forced here by widening it);
    a result outside the search grid is never a tie."""
    import numpy as np
    from oracle_lib import Oracle
    b = _bench()
    cfg = b.CONFIGS[1]
    bits = np.frombuffer(open(os.path.join(ROOT, "tests", "golden", "synth_nott_fs5456.bin"), "rb").read()[:6 * 5120], dtype=np.uint8)
    _, cpu = Oracle(cfg["fc"], cfg["fs"], cfg["max_fo"], kind="f32").bench_blocks(bits, 6)
    _, gpu = Oracle(cfg["fc"], cfg["fs"], cfg["max_fo"], kind="f64").search(bits.tobytes())
    r, _, _ = b.compare_peaks(cfg, gpu, cpu, bits)
    assert r["blocks"] == 6 and r["ca_equal"] == 6 and r["lo_equal"] == 6 and r["n_unproven"] == 0 and r["snr_max_rel"] < 1e-4
    moved = gpu.copy()
    moved["ca_shift"][3] = (moved["ca_shift"][3] + 7) % 5456
    r, _, _ = b.compare_peaks(cfg, moved, cpu, bits)
    assert r["ca_equal"] == 5 and r["unproven_mismatches"] == [3] and r["proven_ties"] == 0
    monkeypatch.setattr(b, "PARITY_TIE_REL", 10.0)  # "anything is a tie": the proven-tie branch itself
    r, _, _ = b.compare_peaks(cfg, moved, cpu, bits)
    assert r["proven_ties"] == 1 and r["n_unproven"] == 0
    off_grid = gpu.copy()
    off_grid["lo_shift"][2] = 500
    r, _, _ = b.compare_peaks(cfg, off_grid, cpu, bits)
    assert r["unproven_mismatches"] == [2]
