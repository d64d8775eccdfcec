#!/usr/bin/env python3
"""bench_extras.py -- everything bench.py reports BESIDE the K timed steps (none of it inside the timed region):

  ClockSampler / cu_clocks      the clock the roofline is quoted at (sysfs readings; the CUs' own cycle counters)
  soak                          the timed step repeated for >= 6 s (an SMI sampler beside the run sees the GPU busy)
  pk_fma_stream                 a pure v_pk_fma_f32 stream: what the fp32 vector pipe sustains on this box
  live_traffic                  roofline.traffic measured in the run (two rocprofv3 --pmc child passes)
  cpu_baseline*, compare_peaks, parity_vs_gpu
                                the oracle (test infrastructure, the CHECKER) timed on the host cores, and the timed step's own results
                                checked against it; a disagreement makes bench.py exit 3
  gpu_library_baseline          the same cells through rocFFT (torch.fft) on the same GPU: the vendor-library figure beside `value`
  inproc_multi                  N > 1: one process drives the N devices through the C ABI's own gpsacq_multi_search_blocks (RCCL via
                                ncclCommInitAll) over the SAME capture; its merged keys must hash to the line's keys_digest
  e2e_cli                       wall clock of the gps_test front end on a file of the bench's size
  strong_share_at_8, one_rank_collective, weak_scaling   further legs of the same step (they reuse bench.Leg)
"""
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "gnss-gps-sdr_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

N_FFT = 40000
STAMP_SLOTS = 512  # gpsacq.h GPSACQ_STAMP_SLOTS
L2_LINE_BYTES = 128  # gfx950: one TCP->TCC read request moves a 128-byte line (the FETCH_SIZE correction of MI355X_MICROARCH.md, one level up)

FP32_VALU_PEAK_TF = 157.3        # MI355X_MICROARCH.md: peak FP32 vector (= FP32 MFMA) rate, dense
FP32_PEAK_CLOCK_MHZ = 2400.0     # the clock that figure assumes: 256 CUs x 128 lanes x 2 flop x 2.4 GHz
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
L2_PEAK_GBS = 34500.0            # MI355X_MICROARCH.md: aggregate L2 bandwidth
ALG_BYTES_PER_CELL = 32 * N_FFT  # read signal + code spectra, write + read one IFFT intermediate
CAPTURE_SEED = 1000              # THE synthetic capture: every rank of every world size generates its own blocks of this one stream

# PRN -> G2 tap pair (c/search_offline.cpp:20-53), for the synthetic IQ capture of --input iq8
PRN_TAPS = [(2, 6), (3, 7), (4, 8), (5, 9), (1, 9), (2, 10), (1, 8), (2, 9), (3, 10), (2, 3), (3, 4), (5, 6), (6, 7), (7, 8), (8, 9), (9, 10),
            (1, 4), (2, 5), (3, 6), (4, 7), (5, 8), (6, 9), (1, 3), (4, 6), (5, 7), (6, 8), (7, 9), (8, 10), (1, 6), (2, 7), (3, 8), (4, 9)]

CONFIGS = {
    1: dict(fc=4.092e6, fs=5.456e6, max_fo=5000.0, name="BASELINE configs[1]"),
    2: dict(fc=2.046e6, fs=8.184e6, max_fo=5000.0, name="BASELINE configs[2]"),
    3: dict(fc=0.62e6, fs=2.8e6, max_fo=100000.0, name="BASELINE configs[3]"),
    4: dict(fc=4.092e6, fs=5.456e6, max_fo=100000.0, name="BASELINE configs[4]"),
}


def flops_per_cell(nlags):
    """6N (conj-multiply) + 5 N log2 N (IFFT-40000) + 5 S (peak scan)."""
    return 6.0 * N_FFT + 5.0 * N_FFT * math.log2(N_FFT) + 5.0 * nlags


def synth_sats(seed, fs):
    """The 8 satellites injected into the capture generated from `seed`: (sorted PRNs, generator tuples)."""
    rs = np.random.default_rng(seed)
    prns = sorted(rs.choice(np.arange(1, 33), size=8, replace=False).tolist())
    return prns, [(prn, 0.151, float(rs.uniform(-4500, 4500)), float(rs.uniform(0, fs / 1000)), float(rs.random())) for prn in prns]


def make_iq_capture(torch, dev, fs, n_blocks, seed):
    """rtl-sdr style capture on the device (torch): uint8 offset-128 interleaved I,Q at BASEBAND -- complex noise of sigma 30 + the
    seeded PRNs at their Doppler + a DC offset; the engine mixes it up to the config's IF (proc_rtl_bin_for_gps.m:31-47) inside the
    forward transform.  Returns (bytes, (mean_i, mean_q))."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    out = torch.empty(n_blocks * 81920, dtype=torch.uint8, device=dev)
    tables = list(synth_sats(seed, fs)[1])
    chip_tab = {}
    for prn, *_ in tables:  # the 1023 chips of each PRN (G1 = x^10+x^3+1, G2 = x^10+x^9+x^8+x^6+x^3+x^2+1, c/cacode.h:9-35)
        g1, g2 = [1] * 10, [1] * 10
        t1, t2 = PRN_TAPS[prn - 1]
        c = []
        for _ in range(1023):
            c.append(g1[9] ^ g2[t1 - 1] ^ g2[t2 - 1])
            g1, g2 = [g1[2] ^ g1[9]] + g1[:9], [g2[1] ^ g2[2] ^ g2[5] ^ g2[7] ^ g2[8] ^ g2[9]] + g2[:9]
        chip_tab[prn] = torch.tensor([1.0 - 2.0 * v for v in c], dtype=torch.float64, device=dev)
    for b0 in range(0, n_blocks, 64):
        nb = min(64, n_blocks - b0)
        m = torch.arange(b0 * 40960, (b0 + nb) * 40960, dtype=torch.float64, device=dev)
        zr = torch.randn(m.numel(), generator=g, device=dev, dtype=torch.float32).double() / math.sqrt(2)
        zi = torch.randn(m.numel(), generator=g, device=dev, dtype=torch.float32).double() / math.sqrt(2)
        for prn, amp, dop, ca, ph in tables:
            idx = torch.floor((m + ca) * (1.023e6 * (1 + dop / 1575.42e6) / fs)).long() % 1023
            th = 2 * math.pi * ((dop / fs * m + ph) % 1.0)
            a = amp * chip_tab[prn][idx]  # rails of sigma 1/sqrt 2 and envelope amp/sqrt 2: amp over a unit-sigma real IF after the mixer
            zr += a * torch.cos(th) / math.sqrt(2)
            zi += a * torch.sin(th) / math.sqrt(2)
        seg = out[b0 * 81920:(b0 + nb) * 81920].view(-1, 2)
        seg[:, 0] = torch.clamp(torch.round(30.0 * zr + 3.7) + 128, 0, 255).to(torch.uint8)
        seg[:, 1] = torch.clamp(torch.round(30.0 * zi + 1.2) + 128, 0, 255).to(torch.uint8)
    sums = out.view(-1, 2).sum(dim=0, dtype=torch.int64).cpu()
    n = out.numel() // 2
    return out, (float(sums[0]) / n - 128.0, float(sums[1]) / n - 128.0)


PARITY_SNR_REL = 1e-4   # BASELINE.json north_star: correlator magnitudes within 1e-4 relative
PARITY_PWR_REL = 2e-5   # what the GPU suite asserts per cell (tests/test_gpu_parity.py REL)
PARITY_SCREEN_REL = 1e-5  # cells: within this of the oracle's FLOAT build passes; further off, the double build judges at PARITY_PWR_REL
PARITY_TIE_REL = 1e-5   # two candidates closer than this in the double-precision oracle are a float-rounding tie


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


def keys_digest(keys):
    """sha256 over the merged keys (int64, little endian) of a step: identical for every world size by construction -- every N
    searches the same capture, and integer MAX is associative."""
    import hashlib
    return hashlib.sha256(np.ascontiguousarray(np.asarray(keys, dtype="<i8")).tobytes()).hexdigest()[:16]


def pack_keys_host(peaks, kmax):
    """gpsacq_peak_keys_device's packing on PEAK_DTYPE records (host): snr bits << 32 | (0xFFFF - (lo + kmax)) << 16 | ca."""
    snr_bits = peaks["snr"].astype("<f4").view("<u4").astype(np.int64)
    lo = peaks["lo_shift"].astype(np.int64) + kmax
    return (snr_bits << 32) | ((0xFFFF - lo) << 16) | (peaks["ca_shift"].astype(np.int64) & 0xFFFF)


def hbm_copy_gbs(torch, dev, nbytes=1 << 30, reps=5):
    """Device-to-device copy rate (read + write bytes per second): the measured counterpart of the 8 TB/s vendor peak."""
    a = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    b = torch.empty_like(a)
    b.copy_(a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        b.copy_(a)
    e1.record()
    e1.synchronize()
    return 2.0 * nbytes * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9


# ---- clock / power -------------------------------------------------------------------------------------------------------
def smi_sample(dev_index=0):
    """One sclk / package power / junction temperature reading (rocm-smi; best effort -- None fields when the tool or a field is missing)."""
    import re
    import subprocess
    out = {"sclk_mhz": None, "power_w": None, "junction_c": None}
    try:
        r = subprocess.run(["rocm-smi", "-d", str(dev_index), "--showpower", "--showclocks", "--showtemp"], capture_output=True, text=True, timeout=20)
        for ln in r.stdout.splitlines():
            m = re.search(r"sclk clock level.*\((\d+)Mhz\)", ln)
            if m:
                out["sclk_mhz"] = int(m.group(1))
            m = re.search(r"Power \(W\):\s*([0-9.]+)", ln)
            if m:
                out["power_w"] = float(m.group(1))
            m = re.search(r"junction\).*:\s*([0-9.]+)", ln)
            if m:
                out["junction_c"] = float(m.group(1))
    except Exception as ex:
        out["error"] = str(ex)[:100]
    return out


def _drm_card(torch, dev_index):
    """sysfs directory /sys/class/drm/cardN/device of the torch device (matched by PCI address when there are several)."""
    import glob
    cands = [d for d in sorted(glob.glob("/sys/class/drm/card[0-9]*/device")) if os.path.exists(os.path.join(d, "pp_dpm_sclk"))]
    if len(cands) <= 1:
        return cands[0] if cands else None
    try:
        pr = torch.cuda.get_device_properties(dev_index)
        bdf = "%04x:%02x:%02x" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        for d in cands:
            if bdf in os.path.realpath(d):
                return d
    except Exception:
        pass
    return cands[min(dev_index, len(cands) - 1)]


def read_clock_power(card):
    """One reading of the shader clock (the level pp_dpm_sclk marks with '*') and the package power (hwmon, microwatts) from sysfs."""
    import glob
    import re
    out = {"sclk_mhz": None, "power_w": None}
    try:
        for ln in open(os.path.join(card, "pp_dpm_sclk")):
            m = re.search(r"(\d+)\s*Mhz\s*\*", ln, re.I)
            if m:
                out["sclk_mhz"] = int(m.group(1))
    except OSError:
        pass
    for name in ("power1_average", "power1_input"):
        for f in glob.glob(os.path.join(card, "hwmon", "hwmon*", name)):
            try:
                out["power_w"] = float(open(f).read().strip()) * 1e-6
                return out
            except (OSError, ValueError):
                pass
    return out


def cu_clocks(stamps, ms):
    """stamps[2][STAMP_SLOTS]: every compute unit's shader-cycle counter before / after a stretch of busy work (gpsacq_cycle_stamp_device;
    slot = xcc << 6 | se << 4 | cu, 0 = not reached), ms: the time between the two stamp kernels.  Each CU's counter has its own offset
    and stands still while the CU is gated, so only same-slot differences count.  Returns (chip MHz = mean of the per-XCD medians,
    [per-XCD median MHz], CUs that gave a reading); (None, None, n) when fewer than half of the XCDs can be read."""
    a, b = stamps[0].astype(np.int64), stamps[1].astype(np.int64)
    ok = (a > 0) & (b > a) & (ms > 0)
    mhz = np.where(ok, (b - a) / max(ms * 1e3, 1e-9), np.nan)
    mhz[(mhz < 300.0) | (mhz > 4000.0)] = np.nan
    per_xcd = []
    for x in range(8):
        v = mhz[64 * x:64 * (x + 1)]
        v = v[~np.isnan(v)]
        per_xcd.append(round(float(np.median(v)), 1) if v.size >= 4 else None)
    good = [m for m in per_xcd if m is not None]
    return (float(np.mean(good)) if len(good) >= 4 else None), (per_xcd if good else None), int(np.count_nonzero(~np.isnan(mhz)))


class ClockSampler:
    """sclk / power readings every `period` seconds from a thread while a leg runs (sysfs; falls back to ONE rocm-smi call in
    mid-leg when sysfs has no clock file).  stats(): medians over the readings taken between start() and stop()."""

    def __init__(self, torch, dev_index, period=0.1):
        import threading
        self.card = _drm_card(torch, dev_index)
        self.dev_index, self.period = dev_index, period
        self.readings, self._stop, self._th = [], threading.Event(), None

    def _loop(self):
        if self.card is None:
            if not self._stop.wait(0.3):
                r = smi_sample(self.dev_index)
                self.readings.append({"sclk_mhz": r.get("sclk_mhz"), "power_w": r.get("power_w"), "t": time.perf_counter()})
            return
        wait = min(0.01, self.period)  # a first reading 10 ms in (a leg of a few steps still gets one), then one per period
        while not self._stop.wait(wait):
            r = read_clock_power(self.card)
            r["t"] = time.perf_counter()
            self.readings.append(r)
            wait = self.period

    def start(self):
        import threading
        self.readings, self._stop = [], threading.Event()
        self._th = threading.Thread(target=self._loop, daemon=True)
        self.t0 = time.perf_counter()
        self._th.start()

    def stop(self):
        self.t1 = time.perf_counter()
        self._stop.set()
        if self._th is not None:
            self._th.join(timeout=30)

    def stats(self):
        inside = [r for r in self.readings if self.t0 <= r["t"] <= self.t1] or self.readings  # (a late rocm-smi reading: better than none)
        clk = [r["sclk_mhz"] for r in inside if r.get("sclk_mhz")]
        pw = [r["power_w"] for r in inside if r.get("power_w")]
        return {"sclk_mhz": float(np.median(clk)) if clk else None, "power_w": float(np.median(pw)) if pw else None,
                "sclk_mhz_min_max": [min(clk), max(clk)] if clk else None, "samples": len(inside),
                "source": "sysfs pp_dpm_sclk / hwmon power1, every %.0f ms during the timed steps" % (1e3 * self.period) if self.card else "rocm-smi, one reading"}


def clock_of_leg(leg):
    """The roofline's clock: the GPU's own cycle counts over the timed steps (s_memtime stamps of every CU / HIP events on the engine's
    stream; mean over the XCDs -- under the power cap they run up to 5 % apart and sysfs follows XCD 0), sysfs readings beside it."""
    clock = leg.sampler.stats()
    clock["sclk_mhz_sysfs"] = clock["sclk_mhz"]
    mt = getattr(leg, "memtime_mhz", None)
    clock["sclk_mhz_cycle_counter"] = mt
    clock["sclk_mhz_per_xcd"] = getattr(leg, "memtime_per_xcd", None)
    clock["cycle_counter_cus"] = getattr(leg, "memtime_cus", None)
    clock["closing_stamp_kernel_ms"] = getattr(leg, "stamp_kernel_ms", None)  # the one launch of the timed region that is not a step
    good = [m for m in (clock["sclk_mhz_per_xcd"] or []) if m]
    clock["sclk_mhz_xcd_min_max"] = [min(good), max(good)] if good else None
    if mt:
        clock["sclk_mhz"] = mt
        clock["source"] = ("shader-cycle counters of every CU (s_memtime stamps around the timed steps / HIP-event time between them): mean of the "
                           "per-XCD medians; sysfs beside it: " + clock["source"])
    return clock


def soak(leg, cells_job, min_gpu_s=6.0, max_steps=2000):
    """The timed step again until >= min_gpu_s of GPU time: steps are enqueued in slices (the wall time of a slice that ends in a
    synchronize), an SMI reading is taken from a thread in mid-leg, the CUs' cycle counters are stamped around the whole leg."""
    import threading
    readings = []
    th = None
    done_s, steps, slices = 0.0, 0, []
    torch = leg.torch
    stamps = torch.zeros((2, STAMP_SLOTS), dtype=torch.int64, device=leg.dev)
    ev = [torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)]
    leg.fence()
    leg.eng.cycle_stamp_device(stamps[0].data_ptr())
    ev[0].record(leg.eng_stream)
    while done_s < min_gpu_s and steps < max_steps:
        n = 20
        leg.fence()
        t0 = time.perf_counter()
        for _ in range(n):
            leg.step()
        if th is None and done_s >= 0.25 * min_gpu_s:  # GPU queue is full for the next ~n steps: read clocks / power now
            th = threading.Thread(target=lambda: readings.append(smi_sample(leg.dev.index or 0)))
            th.start()
        leg.fence()
        dt = time.perf_counter() - t0
        slices.append(1e3 * dt / n)
        done_s += dt
        steps += n
    if th is not None:
        th.join()
    leg.eng.cycle_stamp_device(stamps[1].data_ptr())
    ev[1].record(leg.eng_stream)
    leg.fence()
    chip_mhz, per_xcd, _ = cu_clocks(stamps.cpu().numpy(), ev[0].elapsed_time(ev[1]))
    ms = 1e3 * done_s / max(steps, 1)
    return {"steps": steps, "seconds": done_s, "ms_per_step": ms, "cells_per_s": cells_job / (ms * 1e-3) if ms else None,
            "ms_per_step_slices_min_max": [min(slices), max(slices)] if slices else None,
            "sclk_mhz": chip_mhz, "sclk_mhz_per_xcd": per_xcd, "smi_mid_leg": readings[0] if readings else None,
            "note": "same step as the timed region, repeated after it; reported separately, `steps`/`ms_per_step`/`value` are the K timed steps"}


def pk_fma_stream(seconds=3.0):
    """gnss-gps-sdr_amd/bin/pk_fma_stream (tools/ubench/pk_fma_stream.hip): the rate of a pure v_pk_fma_f32 stream at k_corr's
    residency for `seconds`.  Returns its JSON or {"error"}."""
    import subprocess
    exe = os.path.join(ROOT, "gnss-gps-sdr_amd", "bin", "pk_fma_stream")
    if not os.path.exists(exe):
        return {"error": "pk_fma_stream not built (make host)"}
    try:
        r = subprocess.run([exe, repr(seconds)], capture_output=True, text=True, timeout=60 + 2 * seconds)
        for ln in r.stdout.splitlines():
            if ln.startswith("{"):
                return json.loads(ln)
        return {"error": f"rc {r.returncode}: {(r.stderr or r.stdout)[-200:]}"}
    except Exception as ex:
        return {"error": str(ex)[:200]}


def kernel_source_sha():
    """sha256 over the sources of the timed kernels: profiles/traffic.json carries the value of the tree it was profiled on."""
    import hashlib
    h = hashlib.sha256()
    for name in ("acq_kernels.hip", "acq_phases.hpp", "acq_math.hpp"):
        with open(os.path.join(ROOT, "gnss-gps-sdr_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def live_traffic(argv_tail, timeout_s=150):
    """HBM bytes of one k_corr launch measured IN THIS RUN: two short child runs of bench.py under `rocprofv3 --pmc FETCH_SIZE` and
    `--pmc WRITE_SIZE` (separate passes, as /opt/skills/guides/MI355X_MICROARCH.md prescribes; no trace domains beside --pmc), same
    workload, 2 timed steps.  FETCH_SIZE is in KiB and counts half the bytes of wide coalesced reads on gfx950 (x 2, the guide's
    correction); WRITE_SIZE KiB is uncalibrated.  Returns {"read_bytes", "write_bytes", "launches"} or {"error": ...}; never raises."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None:
        return {"error": "rocprofv3 not on PATH"}
    out = {}
    tmp = tempfile.mkdtemp(prefix="gpsacq_pmc_", dir="/tmp")
    try:
        for counter, key in (("FETCH_SIZE", "read_bytes"), ("WRITE_SIZE", "write_bytes")):
            d = os.path.join(tmp, counter)
            cmd = ["rocprofv3", "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "p", "--", sys.executable, os.path.join(ROOT, "bench.py"),
                   "--steps", "2", "--warmup", "1", "--bare", "--no-dist"] + argv_tail
            env = dict(os.environ, TMPDIR="/tmp")
            for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
                env.pop(k, None)
            r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout_s)
            vals = []
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if "k_corr" in row.get("Kernel_Name", "") and row.get("Counter_Name") == counter:
                        vals.append(float(row["Counter_Value"]))
            if not vals:
                return {"error": f"no {counter} rows for k_corr (rocprofv3 rc {r.returncode}): {(r.stderr or r.stdout)[-200:]}"}
            kib = sum(vals) / len(vals)
            out[key] = kib * 1024.0 * (2.0 if counter == "FETCH_SIZE" else 1.0)
            out["launches"] = len(vals)
        return out
    except Exception as ex:
        return {"error": str(ex)[:200]}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


# ---- the oracle as baseline and as checker ------------------------------------------------------------------------------------
def cpu_baseline(cfg, bits, ndop, target_s=12.0, max_blocks=None):
    """The oracle's float build (own mixed-radix FFT; `port`) timed single-threaded on a bounded sample of the same capture.
    Returns (dict, peaks of exactly those blocks -- what parity_vs_gpu compares the timed step's with)."""
    from oracle_lib import Oracle
    orc = Oracle(cfg["fc"], cfg["fs"], cfg["max_fo"], kind="f32")
    t0 = time.perf_counter()
    cells, _ = orc.bench_blocks(bits[:2 * 5120], 2)
    dt = time.perf_counter() - t0
    nblk = int(max(2, min(len(bits) // 5120, target_s / (dt / 2))))
    if max_blocks:
        nblk = min(nblk, max_blocks)
    t0 = time.perf_counter()
    cells, peaks = orc.bench_blocks(bits[:nblk * 5120], nblk)
    dt = time.perf_counter() - t0
    return ({"value": cells / dt, "unit": "cells/s", "cores": 1, "kind": "port",
             "sample": f"{nblk} blocks x {ndop} bins = {cells} cells of the same capture, oracle f32 build (own FFT, -O3), "
                       f"{dt:.1f} s on {os.cpu_count()} core host ({cpu_model()}), 1 thread",
             "context": "the reference itself needs FFTW3f, absent from this image and from the GPU box (oracle/_ref unbuildable); the survey "
                        "timed the reference's sources against an MKL FFT stand-in at 3.2-5.7 k cells/s per core of a 2.1 GHz Xeon (BASELINE.md "
                        "section 2; SURVEY.md section 8d) -- the same range as this port"}, peaks)


def cpu_baseline_reference(cfg, bits, ndop, target_s=15.0):
    """The reference binary itself (oracle/_ref/gps_test_ref, built by `make -C oracle ref` where a real FFTW3 exists) timed on whole
    runs of the same capture.  None if it was never built."""
    import subprocess
    import tempfile
    exe = os.path.join(ROOT, "oracle", "_ref", "gps_test_ref")
    if not os.path.exists(exe):
        return None
    runs_avail = len(bits) // (32 * 5120)
    if runs_avail < 1:
        return None

    def timed(n_runs):
        with tempfile.NamedTemporaryFile(suffix=".bin") as f:
            f.write(bytes(bits[:n_runs * 32 * 5120]))
            f.flush()
            t0 = time.perf_counter()
            subprocess.run([exe, f.name, repr(cfg["fc"]), repr(cfg["fs"]), "5000"], stdout=subprocess.DEVNULL, check=True)
            return time.perf_counter() - t0
    dt1 = timed(1)
    n = int(max(1, min(runs_avail, target_s / dt1)))
    dt = timed(n) if n > 1 else dt1
    cells = n * 32 * ndop
    return {"value": cells / dt, "unit": "cells/s", "cores": 1, "kind": "reference",
            "sample": f"oracle/_ref/gps_test_ref (the reference's sources + FFTW3f) on {n} runs x 32 PRN x {ndop} bins = {cells} cells of the "
                      f"same capture, {dt:.1f} s incl. process start, on {os.cpu_count()} core host ({cpu_model()}), 1 thread"}


def cpu_baseline_all_cores(cfg, bits, ndop, one_thread_rate, cap_s=60.0, first_block=0):
    """The same port on every core the process may use -- ONE pass over the whole share of the capture this rank searched, inside
    liboracle_f32.so (oracle_search_omp: blocks handed out in ascending order, reference schedule block -> PRN) -- which KEEPS what it
    computes: the all-cores figure and the checker's side of parity_vs_gpu are the same work.  cap_s bounds the leg on a slow host
    (blocks not reached by then are simply not part of the verdict).  Returns (dict, peaks, cells[block][bin], blocks done)."""
    import ctypes
    from oracle_lib import CELL_DTYPE, PEAK_DTYPE, lib
    L = lib("f32")
    L.oracle_search_omp.restype = ctypes.c_long
    L.oracle_search_omp.argtypes = [ctypes.c_double] * 3 + [ctypes.c_void_p, ctypes.c_long, ctypes.c_long, ctypes.c_long, ctypes.c_int, ctypes.c_double,
                                                             ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)]
    try:
        ncpu = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        ncpu = os.cpu_count() or 1
    quota = None  # a container's CPU bandwidth limit (cgroup v2 cpu.max): the cores the process can really use at once
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        quota = None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        pass
    buf = np.ascontiguousarray(bits)
    nblk = buf.size // 5120
    peaks, cells, done = np.zeros(nblk, PEAK_DTYPE), np.zeros((nblk, ndop), CELL_DTYPE), np.zeros(nblk, np.uint8)
    el, used = ctypes.c_double(), ctypes.c_int()
    nthreads = ncpu if quota is None else max(1, min(ncpu, int(math.ceil(quota))))
    n_cells = L.oracle_search_omp(cfg["fc"], cfg["fs"], cfg["max_fo"], buf.ctypes.data, nblk, 5120, int(first_block), nthreads, float(cap_s),
                                  peaks.ctypes.data, cells.ctypes.data, done.ctypes.data, ctypes.byref(el), ctypes.byref(used))
    n_done = int(done.sum())
    if not done[:n_done].all():  # (ascending hand-out and every block taken is finished: the blocks done are a prefix)
        raise RuntimeError("oracle_search_omp: the blocks done are not a prefix of the share")
    rate = n_cells / el.value if el.value > 0 else 0.0
    return ({"value": rate, "unit": "cells/s", "cores": used.value, "kind": "port",
             "sched_getaffinity_cores": ncpu, "os_cpu_count": os.cpu_count(), "cgroup_cpu_quota_cores": quota,
             "speedup_over_1_thread": rate / one_thread_rate if one_thread_rate else None, "blocks": n_done, "blocks_of_the_share": nblk, "seconds": el.value,
             "sample": f"OpenMP, {used.value} threads (sched_getaffinity: {ncpu} cores, cgroup cpu.max quota: {quota} cores), one pass over {n_done} of the {nblk} blocks "
                       f"this rank searched x {ndop} bins = {n_cells} cells in {el.value:.1f} s; its peaks and cells are what parity_vs_gpu compares the timed step's with"},
            peaks[:n_done], cells[:n_done], n_done)


def compare_peaks(cfg, gpu_peaks, cpu_peaks, host_bits, first_block=0):
    """Part (1) of parity_vs_gpu, host only: the GPU's peaks against the oracle's for the same blocks of the same capture (reference
    schedule: block b against PRN b % 32; first_block: the capture index of block 0 of host_bits, a multiple of 32).  A GPU value that
    is not finite, or outside the grid, is an unproven mismatch -- never skipped.  Returns (dict, oracle_f64, lag_powers)."""
    from oracle_lib import Oracle, _p
    n = min(len(cpu_peaks), len(gpu_peaks))
    g, o = gpu_peaks[:n], cpu_peaks[:n]
    orc = Oracle(cfg["fc"], cfg["fs"], cfg["max_fo"], kind="f64")
    S = orc.num_lags

    def lag_powers(block_bytes, sv, lo):
        orc.L.oracle_sample(orc.h, _p(np.ascontiguousarray(block_bytes)))
        pw = np.zeros(S, np.float32)
        orc.L.oracle_cell_power(orc.h, sv, int(lo), _p(pw))
        return pw

    ca_eq, lo_eq = g["ca_shift"] == o["ca_shift"], g["lo_shift"] == o["lo_shift"]
    finite = np.isfinite(g["snr"].astype(np.float64))
    ties, unproven = 0, []
    for b in np.nonzero(~(ca_eq & lo_eq & finite))[0]:
        if not finite[b]:
            unproven.append(int(b))
            continue
        blk = np.frombuffer(host_bits, dtype=np.uint8)[b * 5120:(b + 1) * 5120]
        snr2 = []
        for pk in (g[b], o[b]):
            if not (-orc.dmax <= int(pk["lo_shift"]) <= orc.dmax and 0 <= int(pk["ca_shift"]) < S):
                snr2.append(float("nan"))  # a result outside the search grid is never a tie
                continue
            pw = lag_powers(blk, (first_block + int(b)) % 32, pk["lo_shift"])
            snr2.append(float(pw[pk["ca_shift"]]) / (float(pw.sum(dtype=np.float64)) / S))
        if abs(snr2[0] - snr2[1]) <= PARITY_TIE_REL * snr2[1]:
            ties += 1
        else:
            unproven.append(int(b))
    with np.errstate(divide="ignore", invalid="ignore"):
        snr_rel = np.abs(g["snr"].astype(np.float64) / o["snr"].astype(np.float64) - 1.0)
    snr_rel = np.where(np.isfinite(snr_rel), snr_rel, np.inf)  # NaN / inf on either side: not within any tolerance
    snr_max_rel = float(np.max(snr_rel)) if n else 0.0
    return ({"blocks": int(n), "ca_equal": int(ca_eq.sum()), "lo_equal": int(lo_eq.sum()), "proven_ties": ties, "unproven_mismatches": unproven[:8],
             "n_unproven": len(unproven), "snr_max_rel": snr_max_rel, "non_finite_gpu_peaks": int((~finite).sum())}, orc, lag_powers)


def compare_cells(gpu_cells, cpu_cells, host_bits, first_block, orc, lag_powers, max_recheck=20000):
    """Part (2) of parity_vs_gpu, host only: EVERY cell of the blocks the oracle searched (c/search_offline.cpp:190-196: max_pwr, its lag,
    tot_pwr).  Two tiers: the oracle's float build is the screen -- a power within PARITY_SCREEN_REL of it passes --, its double build
    the judge: a power further off, and every cell whose lag differs, is recomputed in double; the power must then be within
    PARITY_PWR_REL, the lag's power within PARITY_TIE_REL of the largest (a float-rounding tie).  Non-finite or non-positive GPU powers
    and lags outside the scan fail outright."""
    n = min(len(gpu_cells), len(cpu_cells))
    g, o = gpu_cells[:n], cpu_cells[:n]
    dmax, S = orc.dmax, orc.num_lags
    bad_values, worst_screen, judge = 0, 0.0, {}
    for f in ("max_pwr", "tot_pwr"):
        gv, ov = g[f].astype(np.float64), o[f].astype(np.float64)
        ok = np.isfinite(gv) & (gv > 0)
        bad_values += int((~ok).sum())
        with np.errstate(divide="ignore", invalid="ignore"):
            rel = np.where(ok & (ov > 0), np.abs(gv / ov - 1.0), np.inf)
        over = rel > PARITY_SCREEN_REL
        if (~over).any():
            worst_screen = max(worst_screen, float(rel[~over].max()))
        for b, d in zip(*np.nonzero(over & ok)):
            judge.setdefault((int(b), int(d)), set()).add(f)
    lag_in = (g["max_i"] >= 0) & (g["max_i"] < S)
    lag_bad = [[first_block + int(b), int(d) - dmax] for b, d in zip(*np.nonzero(~lag_in))]
    for b, d in zip(*np.nonzero((g["max_i"] != o["max_i"]) & lag_in)):
        judge.setdefault((int(b), int(d)), set()).add("lag")
    worst_judged, lag_ties, too_many = 0.0, 0, len(judge) > max_recheck
    pwr_bad = []
    if not too_many:
        bits = np.frombuffer(host_bits, dtype=np.uint8)
        for (b, d), what in sorted(judge.items()):
            pw = lag_powers(bits[b * 5120:(b + 1) * 5120], (first_block + b) % 32, d - dmax).astype(np.float64)
            truth = {"max_pwr": float(pw.max()), "tot_pwr": float(pw.sum())}
            for f in what - {"lag"}:
                rel = abs(float(g[f][b, d]) / truth[f] - 1.0) if truth[f] > 0 else float("inf")
                worst_judged = max(worst_judged, rel)
                if not rel <= PARITY_PWR_REL:
                    pwr_bad.append([first_block + b, d - dmax, f, rel])
            if "lag" in what:
                if truth["max_pwr"] - float(pw[int(g["max_i"][b, d])]) <= PARITY_TIE_REL * truth["max_pwr"]:
                    lag_ties += 1
                else:
                    lag_bad.append([first_block + b, d - dmax])
    ok = bad_values == 0 and not lag_bad and not pwr_bad and not too_many
    return {"ok": bool(ok), "cells": int(g.size), "pwr_max_rel": max(worst_screen, worst_judged), "pwr_max_rel_vs_float_oracle_where_screened": worst_screen,
            "cells_judged_in_double": len(judge), "pwr_max_rel_judged_in_double": worst_judged, "too_many_cells_off_the_screen": bool(too_many),
            "non_finite_or_non_positive_cell_powers": bad_values, "cell_lag_ties": lag_ties, "cell_lag_mismatches": lag_bad[:8], "n_cell_lag_mismatches": len(lag_bad),
            "cell_power_mismatches": pwr_bad[:8], "n_cell_power_mismatches": len(pwr_bad)}


def parity_vs_gpu(cfg, gpu_peaks, gpu_cells, cpu_peaks, cpu_cells, host_bits, first_block=0, share_blocks=None):
    """The timed step's own results against the oracle (test infrastructure; c/search_offline.cpp:190-198,248), AFTER the timed
    region, host only -- the GPU's peaks and cells are the copies taken right behind the last timed step: (1) the peak of every block
    the oracle searched -- same capture, same blocks, reference schedule block -> PRN block % 32 -- must carry the same ca_shift and
    lo_shift and an SNR within 1e-4; a different (lo, ca) is accepted only as a PROVEN tie: in the double-precision oracle the two
    candidates' SNRs agree to 1e-5.  (2) every cell of those blocks, all Doppler bins (compare_cells): powers within 1e-5 of the oracle's
    float build or within 2e-5 of its double build (finite, positive), the lag identical or a tie proven in double."""
    part1, orc, lag_powers = compare_peaks(cfg, gpu_peaks, cpu_peaks, host_bits, first_block)
    n = part1["blocks"]
    if gpu_cells is not None and cpu_cells is not None:
        part2 = compare_cells(gpu_cells[:n], cpu_cells[:n], host_bits, first_block, orc, lag_powers)
    else:
        part2 = {"ok": True, "cells": 0, "pwr_max_rel": 0.0, "note": "no cells compared"}
    ok = part1["n_unproven"] == 0 and part1["snr_max_rel"] <= PARITY_SNR_REL and part2["ok"] and part2["pwr_max_rel"] <= PARITY_PWR_REL
    out = {"ok": bool(ok), "blocks": n, "ca_equal": part1["ca_equal"], "lo_equal": part1["lo_equal"], "proven_ties": part1["proven_ties"],
           "unproven_mismatches": part1["unproven_mismatches"], "snr_max_rel": part1["snr_max_rel"], "non_finite_gpu_peaks": part1["non_finite_gpu_peaks"]}
    out.update({k: v for k, v in part2.items() if k != "ok"})
    out.update({"first_block_of_this_rank": first_block, "blocks_of_this_ranks_share": share_blocks,
                "whole_share": (share_blocks is not None and n == share_blocks),
                "tolerances": {"snr_rel": PARITY_SNR_REL, "pwr_rel": PARITY_PWR_REL, "pwr_screen_rel": PARITY_SCREEN_REL, "tie_rel": PARITY_TIE_REL},
                "what": "peaks AND cells of the LAST TIMED STEP (rank 0's share) vs the oracle on the same blocks: ca_shift / lo_shift equal or a tie "
                        "proven in the double-precision oracle, SNR to 1e-4; every cell's max_pwr / tot_pwr within 1e-5 of the oracle's float build or "
                        "2e-5 of its double build, its lag identical or a tie proven in double"})
    return out


# ---- the same cells through the vendor FFT on the same GPU --------------------------------------------------------------------
def gpu_library_baseline(torch, eng, dev, d_bits, stride, nblk, gpu_peaks=None, n_blocks=32, reps=3):
    """The same cells through rocFFT via torch.fft -- nothing of the reference, nothing of the oracle: for `n_blocks` blocks of the
    capture (block b against PRN b % 32) and every Doppler bin, conj(D) * roll(C, d) -> batched 40000-point c2c inverse (unnormalised) ->
    |.|^2 over the first S lags -> max / argmax / sum (c/search_offline.cpp:181-196; the arithmetic of
    tests/test_gpu_parity.py::test_cells_vs_torch_fp32_reference).  D and C are the engine's own spectra (parity probes), resident
    before anything is timed, like k_corr's inputs.  Three stages timed with HIP events: the products, the library transform, the scan."""
    N, S, ndop, dmax = N_FFT, eng.num_lags, eng.num_doppler, eng.dmax
    nb = int(min(n_blocks, nblk))
    host = d_bits[:(nb - 1) * stride + 5120].cpu().numpy()
    D = torch.from_numpy(np.stack([eng.sample_spectrum(host[b * stride:b * stride + 5120]) for b in range(nb)])).to(dev)
    C = torch.from_numpy(np.stack([eng.code_spectrum(b % 32) for b in range(nb)])).to(dev)
    Dc = torch.conj(D).resolve_conj().contiguous()
    C2 = torch.cat([C, C], dim=1)  # roll(C, d)[i] = C[(i - d) mod N] = C2[(N - d) mod N + i]
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    times = []
    res = None
    for rep in range(reps + 1):
        ev[0].record()
        prod = torch.empty((nb, ndop, N), dtype=torch.complex64, device=dev)
        for k, d in enumerate(range(-dmax, dmax + 1)):
            o = (N - d) % N
            torch.mul(Dc, C2[:, o:o + N], out=prod[:, k, :])
        ev[1].record()
        y = torch.fft.ifft(prod, dim=-1, norm="forward")  # backward transform without the 1/N
        ev[2].record()
        yr = torch.view_as_real(y[..., :S])
        pwr = yr[..., 0] * yr[..., 0] + yr[..., 1] * yr[..., 1]
        mx, mi = pwr.max(dim=-1)
        tot = pwr.sum(dim=-1)
        snr = mx / (tot / S)
        best_snr, best_k = snr.max(dim=-1)
        ev[3].record()
        ev[3].synchronize()
        if rep > 0:  # (the first pass plans the transform and warms the allocator)
            times.append([ev[i].elapsed_time(ev[i + 1]) for i in range(3)])
        res = (best_snr.cpu().numpy(), (best_k - dmax).cpu().numpy(), mi.gather(1, best_k[:, None]).flatten().cpu().numpy())
        del prod, y, yr, pwr
    t = np.mean(np.asarray(times), axis=0)
    cells = nb * ndop
    out = {"library": "rocFFT through torch.fft.ifft (complex64, batched 40000-point c2c, norm='forward')", "cells": cells,
           "ms_product": float(t[0]), "ms_ifft": float(t[1]), "ms_scan": float(t[2]),
           "cells_per_s": cells / (float(t.sum()) * 1e-3), "cells_per_s_ifft_only": cells / (float(t[1]) * 1e-3),
           "hbm_bytes_per_cell_this_path": 8 * N * 5 + 8 * S,  # product written, read by the transform, transform written (out of place), read by the scan; inputs 2 x 8N
           "note": "same GPU, same cells, inputs resident; product + transform + scan as separate library / element-wise kernels with the "
                   "40000-point intermediates in HBM -- what k_corr's fusion avoids.  Not the reference (FFTW on a CPU) and not a bound."}
    if gpu_peaks is not None and len(gpu_peaks) >= nb:
        g = gpu_peaks[:nb]
        strong = g["snr"] >= 25  # noise-only peaks may tie within float rounding; detections may not
        out["agrees_with_k_corr"] = {"blocks": nb, "detections": int(strong.sum()),
                                     "ca_equal_on_detections": int((res[2][strong] == g["ca_shift"][strong]).sum()),
                                     "lo_equal_on_detections": int((res[1][strong] == g["lo_shift"][strong]).sum()),
                                     "snr_max_rel": float(np.max(np.abs(res[0] / g["snr"] - 1.0)))}
    return out


# ---- N > 1: the C ABI's own multi-GPU path over the same capture --------------------------------------------------------------
def inproc_multi_child(argv):
    """Child process of inproc_multi (its own HIP context): generates the capture of `--runs` runs on device 0 with the library's
    generator, hands it to gpsacq_multi_search_blocks over `--devices` as a host buffer, prints {"ms", "keys_digest", ...}."""
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--fc", type=float)
    ap.add_argument("--fs", type=float)
    ap.add_argument("--max-fo", type=float)
    ap.add_argument("--runs", type=int)
    ap.add_argument("--seed", type=int)
    ap.add_argument("--devices")
    a = ap.parse_args(argv)
    import gpsacq
    from bench import synth_sats
    devices = [int(d) for d in a.devices.split(",")]
    with gpsacq.Engine(a.fc, a.fs, a.max_fo, device=devices[0]) as eng:
        bits = eng.generate(a.runs * 32 * 5120, synth_sats(a.seed, a.fs)[1], noise_sigma=1.0, seed=a.seed)
    with gpsacq.MultiEngine(a.fc, a.fs, a.max_fo, devices=devices) as m:
        m.search_blocks(bits)  # warm-up: scratch, communicator
        t0 = time.perf_counter()
        peaks, best = m.search_blocks(bits)
        ms = 1e3 * (time.perf_counter() - t0)
        call = m.last_call_ms()
        keys = pack_keys_host(best, m.kmax)
    print(json.dumps({"ms": ms, "keys_digest": keys_digest(keys), "devices": devices, "distinct_devices": len(set(devices)),
                      "rccl_allreduces": call["rccl_allreduces"], "enqueue_ms": call["enqueue_ms"], "runs": a.runs,
                      "cells_per_s": a.runs * 32 * (2 * m.kmax + 1) / (ms * 1e-3)}))


def inproc_multi(cfg, total_runs, seed, devices, want_digest, timeout_s=120):
    """After the timed region at N > 1, from rank 0: ONE process drives all N devices through gpsacq_multi_search_blocks -- the C
    ABI's own decomposition (one engine per device, ncclCommInitAll, ONE ncclAllReduce(MAX) of 32 keys) -- over the capture the ranks
    just searched.  Runs as a child process with a time limit (a hang there must not take the bench line with it); the other ranks'
    processes are idle at a barrier meanwhile.  Its merged keys must hash to the line's keys_digest."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "inproc_multi", "--fc", repr(cfg["fc"]), "--fs", repr(cfg["fs"]), "--max-fo", repr(cfg["max_fo"]),
           "--runs", str(total_runs), "--seed", str(seed), "--devices", ",".join(str(d) for d in devices)]
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK", "LOCAL_WORLD_SIZE", "TORCHELASTIC_RUN_ID"):
        env.pop(k, None)
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, env=env)
        for ln in r.stdout.splitlines():
            if ln.startswith("{"):
                j = json.loads(ln)
                j["keys_equal_digest"] = j.get("keys_digest") == want_digest
                return j
        return {"error": f"rc {r.returncode}: {(r.stderr or r.stdout)[-300:]}"}
    except subprocess.TimeoutExpired:
        return {"error": f"timed out after {timeout_s} s"}
    except Exception as ex:
        return {"error": str(ex)[:300]}


# ---- the front end ------------------------------------------------------------------------------------------------------------
def e2e_cli(cfg, d_bits, n_runs, ndop, reps=5):
    """The drop-in a user runs: wall clock of gnss-gps-sdr_amd/bin/gps_test on a capture FILE of the bench's size (written from the
    resident synthetic capture), process start to exit, with the front end's own split (GPSACQ_TRACE)."""
    import re
    import subprocess
    import tempfile
    exe = os.path.join(ROOT, "gnss-gps-sdr_amd", "bin", "gps_test")
    if not os.path.exists(exe):
        return {"error": "gps_test not built"}
    host = d_bits[:n_runs * 32 * 5120].cpu().numpy()
    tmpdir = "/dev/shm" if os.path.isdir("/dev/shm") else None
    with tempfile.NamedTemporaryFile(suffix=".bin", dir=tmpdir) as f:
        host.tofile(f)
        f.flush()
        walls, traces, runs = [], [], 0
        for _ in range(reps):
            t0 = time.perf_counter()
            r = subprocess.run([exe, f.name, repr(cfg["fc"]), repr(cfg["fs"]), "5000"], capture_output=True, text=True,
                               env=dict(os.environ, GPSACQ_TRACE="1"))
            walls.append(time.perf_counter() - t0)
            if r.returncode != 0:
                return {"error": f"gps_test exit {r.returncode}: {r.stderr[-300:]}"}
            runs = r.stdout.count("satellite:")
            traces.append([ln for ln in r.stderr.splitlines() if ln.startswith("gpsacq trace")][-1])
    floor_exe, floor = os.path.join(ROOT, "gnss-gps-sdr_amd", "bin", "hip_floor"), None
    if os.path.exists(floor_exe):  # a HIP process that creates a stream, launches an empty kernel and exits
        fw = []
        for _ in range(reps):
            t0 = time.perf_counter()
            subprocess.run([floor_exe], capture_output=True)
            fw.append(time.perf_counter() - t0)
        floor = min(fw)
    best = int(np.argmin(walls))
    nums = {k: float(v) for k, v in re.findall(r"(SearchInit|SearchTask|mean pass|buffers|read|submit|wait for GPU|report) ([0-9.]+)", traces[best])}
    cells = runs * 32 * ndop
    return {"wall_s": walls[best], "wall_s_median": float(np.median(walls)), "wall_s_all": walls, "runs_reported": runs, "cells": cells, "cells_per_s": cells / walls[best],
            "file_bytes": int(host.size), "split_ms": nums, "hip_process_floor_s": floor,
            "wall_above_floor_s": (walls[best] - floor) if floor else None,
            "note": "process start + HIP runtime/module load + SearchInit + pipelined SearchTask (fread k+1 || search k || printf k-1)"}


# ---- further legs of the same step (bench.Leg) --------------------------------------------------------------------------------
def strong_share_at_8(Leg, base, elapsed_per_step_ms, steps):
    """What one GPU's share of the capture costs at N = 8 (43 of the 340 runs, the one-rank all-reduce included): a PREDICTION of the
    strong-scaling point the driver measures on an 8-GPU node, from this one GPU."""
    share_runs = -(-(base.nblk // 32) // 8)
    sleg = Leg(base.torch, base.eng, base.dev, base.dist, base.backend, share_runs * 32, share_runs * 32, base.d_bits, None, base.stride, False)
    n = max(20, steps // 2)
    s_elapsed, s_kern_ms, _ = sleg.run(n, 3)
    s_ms = 1e3 * s_elapsed / n
    s_tm = base.eng.last_timing()
    return {"ranks_emulated": 8, "blocks_per_step": share_runs * 32, "ms_per_step": s_ms, "kernel_ms": s_kern_ms,
            "stage_ms": {k: s_tm[k] for k in ("ms_sample", "ms_correlate", "ms_peaks")},
            "non_kernel_ms": s_ms - (s_tm["ms_sample"] + s_tm["ms_correlate"] + s_tm["ms_peaks"]),
            "cells_per_s_this_gpu": share_runs * 32 * base.eng.num_doppler / (s_ms * 1e-3),
            "predicted_speedup_at_8_gpus": elapsed_per_step_ms / s_ms,
            "note": "one GPU running the largest per-rank share of the capture at N = 8 (same step, one-rank all-reduce included); "
                    "a prediction, not a measurement of 8 GPUs"}


def one_rank_collective(Leg, base, elapsed_per_step_ms, steps):
    """What the one-rank process group (and the clock sampler / cycle stamps of the timed leg) cost the N = 1 line: the same step
    without any of them."""
    nleg = Leg(base.torch, base.eng, base.dev, None, base.backend, base.nblk, base.n_tasks, base.d_bits, base.d_tasks, base.stride, base.grid,
               n_keys=base.n_keys, iq=base.iq)
    n = max(10, steps // 2)
    n_el, n_kern, _ = nleg.run(n, 2)
    return {"ms_per_step_without_process_group": 1e3 * n_el / n, "kernel_ms": n_kern, "ms_per_step_with": elapsed_per_step_ms}


def roofline_colimiters(onchip, cyc_cell_cu, kern_ms, cells_rank, power_w, l2_peak_gbs):
    """The three pipes k_corr keeps busy at once, from the committed PMC passes (profiles/traffic.json `onchip_counters`) and this run's
    time: VALU (wave-instructions x 4 cycles / 4 SIMDs), LDS (SQ_LDS_IDX_ACTIVE cycles), L2 -> L1 (TCP_TCC_READ_REQ x 128-byte lines
    against the 34.5 TB/s aggregate); and the energy per cell at the package power read during the steps."""
    out = {}
    if onchip and cyc_cell_cu:
        if onchip.get("valu_wave_instr_per_cell"):
            out["valu_busy_frac"] = onchip["valu_wave_instr_per_cell"] / 4.0 * 4.0 / cyc_cell_cu
        if onchip.get("lds_active_cycles_per_cell"):
            out["lds_frac"] = onchip["lds_active_cycles_per_cell"] / cyc_cell_cu
    if onchip and onchip.get("l2_to_l1_bytes_per_cell") and kern_ms and cells_rank:
        gbs = onchip["l2_to_l1_bytes_per_cell"] * cells_rank / (kern_ms * 1e-3) / 1e9
        out["l2_to_l1_GBs"] = gbs
        out["l2_frac"] = gbs / l2_peak_gbs
        out["l2_to_l1_bytes_per_cell"] = onchip["l2_to_l1_bytes_per_cell"]
    if power_w and kern_ms and cells_rank:
        out["energy_uj_per_cell"] = power_w * kern_ms * 1e-3 / cells_rank * 1e6
    return out


def build_line(c):
    """Rank 0, after the timed region: the JSON line from what main() measured (`c`: its locals) plus the untimed legs -- the oracle as
    baseline (N = 1) and as checker of rank 0's share (any N), the rocFFT baseline, the in-process multi-GPU leg, the front end.
    Returns (dict, parity_failed)."""
    parity_failed = False
    fl = flops_per_cell(c.eng.num_lags)
    achieved_tf = c.cells_rank * fl / (c.kern_ms * 1e-3) / 1e12 if c.kern_ms else 0.0
    traffic, traffic_src, onchip, traffic_sha = None, None, None, None
    try:  # HBM bytes per launch and pipe utilisation from the committed PMC passes (profiles/traffic.json)
        tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        traffic = (tj["hbm_read_bytes_per_cell"] + tj["hbm_write_bytes_per_cell"]) * c.cells_rank
        traffic_src = f"profiles/{tj['tag']}_summary.md (rocprofv3 --pmc FETCH_SIZE, WRITE_SIZE in separate passes; bytes per cell x cells per launch)"
        onchip, traffic_sha = tj.get("onchip_counters"), tj.get("kernel_source_sha")
    except Exception:
        pass
    traffic_live = None
    if (c.world == 1 and not c.args.no_live_traffic and c.args.config == 1 and not c.iq8 and not c.args.capture
            and not any(k.startswith(("ROCP_", "ROCPROF")) for k in os.environ) and "rocprof" not in os.environ.get("LD_PRELOAD", "")):  # not under a profiler already
        traffic_live = live_traffic(["--blocks-total", str(c.args.blocks_total)])  # counters collected on THIS box, in this run
        if "error" not in traffic_live:
            traffic = traffic_live["read_bytes"] + traffic_live["write_bytes"]
            traffic_src = ("live (estimated correction): rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate child runs of this command, 2 steps each), "
                           "per k_corr launch; FETCH_SIZE KiB x 1024 x 2 (the guide's gfx950 correction for wide reads), WRITE_SIZE KiB x 1024 (uncalibrated)")
    sha_now = kernel_source_sha()
    live_ok = traffic_live is not None and "error" not in traffic_live
    # counters belong to the kernel binary they were collected on: flagged when the sources changed since, or when this run's instance is not the profiled one
    traffic_stale = False if live_ok else ((traffic is not None) and (traffic_sha != sha_now or c.args.config not in (1, 4) or c.iq8))
    onchip_stale = (onchip is not None) and (traffic_sha != sha_now)
    alg_gbs = c.cells_rank * ALG_BYTES_PER_CELL / (c.kern_ms * 1e-3) / 1e9 if c.kern_ms else 0.0
    # box-independent forms of the same measurement: cycles one CU spends per cell at the clock held DURING the timed steps
    sclk = c.clock["sclk_mhz"] if c.clock else None
    cyc_cell_cu = (c.kern_ms * 1e-3 * sclk * 1e6 * c.eng.compute_units / c.cells_rank) if (sclk and c.kern_ms and c.cells_rank) else None
    frac_at_clock = (achieved_tf / (FP32_VALU_PEAK_TF * sclk / FP32_PEAK_CLOCK_MHZ)) if (sclk and achieved_tf) else None
    pk_tf = c.pk_stream.get("pk_fma_stream_TF") if c.pk_stream else None
    colim = roofline_colimiters(onchip if c.args.config in (1, 4) and not c.iq8 else None, cyc_cell_cu, c.kern_ms, c.cells_rank,
                                  c.clock["power_w"] if c.clock else None, L2_PEAK_GBS)
    out = {
        "metric": "(PRN,Doppler) correlation cells/s, 32 PRN @ fs=5.456 MHz" if c.args.config in (1, 4) else f"(PRN,Doppler) correlation cells/s, 32 PRN @ fs={c.fs / 1e6:.3f} MHz",
        "value": c.cells_job * c.args.steps / c.elapsed, "unit": "cells/s", "n_gpus": c.world, "steps": c.args.steps, "warmup": c.args.warmup,
        "ms_per_step": c.ms_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
        "data": "capture file" if c.args.capture else "synthetic",
        "data_detail": (f"file {c.args.capture}" if c.args.capture else
                        (f"device-generated 1-bit real-IF capture, seed {c.data_seed}: white noise + PRNs {c.injected} at 45 dB-Hz, seeded Doppler / code phase; "
                         "every rank generates its own blocks of this ONE stream") if c.args.data == "signals" else "uniform random bits (sign of white noise)"),
        "config": {"workload": c.workload, "fs_hz": c.fs, "if_hz": c.cfg["fc"], "blocks_rank0": c.nblk, "first_block_rank0": c.first_run * 32,
                   "cells_per_step_rank0": c.cells_rank, "cells_per_step_job": c.cells_job, "parallelism": c.parallelism,
                   "input": "8-bit IQ (uint8 offset 128), converted inside the forward transform" if c.iq8 else "1-bit real IF"},
        "rccl_ranks_seen": c.ranks_seen, "dist_backend": (c.backend if c.dist is not None else None), "dist_note": c.dist_note,
        ("tasks_per_rank" if c.grid else "blocks_per_rank"): c.blocks_per_rank, "devices_per_rank": c.devices,
        # What binds k_corr is the fp32 vector pipe, not HBM: the fused kernel keeps the IFFT intermediate in LDS and reads both spectra
        # from L2, so the algorithmic bytes of SURVEY 8(d) never reach HBM; VALU, LDS and L2 -> L1 are busy AT ONCE (co-limiters below).
        "roofline": {"bound": "valu_fp32", "kernel": f"k_corr<{c.eng.acc_columns}>",
                     "cell_handout": ("persistent workgroups draw (task, Doppler point) tickets at run time (gpsacq_set_cell_handout)"
                                      if c.eng.cell_handout and not (c.eng.acc_columns == 40) else "one workgroup per cell"),
                     "achieved": achieved_tf, "peak": FP32_VALU_PEAK_TF, "unit": "TFLOP/s",
                     "frac": achieved_tf / FP32_VALU_PEAK_TF, "flops_per_cell": fl, "flops_definition": "SURVEY.md 8(d): 6N + 5N log2 N + 5S",
                     "kernel_ms": c.kern_ms, "cells_per_launch": c.cells_rank, "kernel_cells_per_s": c.cells_rank / (c.kern_ms * 1e-3) if c.kern_ms else None,
                     "sclk_mhz": sclk, "sclk_mhz_xcd_min": (c.clock.get("sclk_mhz_xcd_min_max") or [None])[0] if c.clock else None,
                     "sclk_mhz_xcd_max": (c.clock.get("sclk_mhz_xcd_min_max") or [None, None])[1] if c.clock else None,
                     "power_w": c.clock["power_w"] if c.clock else None, "compute_units": c.eng.compute_units,
                     "cycles_per_cell_per_cu": cyc_cell_cu, "frac_at_clock": frac_at_clock, "peak_clock_mhz": FP32_PEAK_CLOCK_MHZ,
                     # the co-limiters: share of the CU's cycles the VALU / the LDS array are busy, share of the L2's bandwidth in use
                     "valu_busy_frac": colim.get("valu_busy_frac"), "lds_frac": colim.get("lds_frac"), "l2_frac": colim.get("l2_frac"),
                     "l2_to_l1_GBs": colim.get("l2_to_l1_GBs"), "l2_to_l1_bytes_per_cell": colim.get("l2_to_l1_bytes_per_cell"),
                     "energy_uj_per_cell": colim.get("energy_uj_per_cell"), "colimiters_stale": onchip_stale,
                     "colimiters_source": "profiles/traffic.json onchip_counters (SQ_INSTS_VALU x 4 cycles / 4 SIMDs, SQ_LDS_IDX_ACTIVE, TCP_TCC_READ_REQ x 128 B) over this run's cycles per cell; power read during the steps",
                     "pk_fma_stream_TF": pk_tf, "frac_of_pk_fma_stream": (achieved_tf / pk_tf) if pk_tf else None,
                     "pk_fma_stream_sclk_mhz": c.pk_clock["sclk_mhz"] if c.pk_clock else None, "clock_sampling": c.clock, "pk_fma_stream": c.pk_stream,
                     "traffic": traffic, "traffic_source": traffic_src, "traffic_stale": traffic_stale, "traffic_live": traffic_live,
                     "kernel_source_sha": sha_now, "traffic_kernel_source_sha": traffic_sha,
                     "hbm_secondary": {"algorithmic_bytes_per_cell": ALG_BYTES_PER_CELL, "algorithmic_GBs": alg_gbs, "hbm_peak_GBs": HBM_PEAK_GBS,
                                       "algorithmic_over_hbm_peak": alg_gbs / HBM_PEAK_GBS,
                                       "measured_hbm_GBs": (traffic / (c.kern_ms * 1e-3) / 1e9) if (traffic and c.kern_ms) else None,
                                       "note": "not a bound: the algorithmic bytes stay on chip; measured HBM traffic is what `traffic` reports"},
                     "onchip_counters": onchip, "onchip_counters_stale": onchip_stale, "l2_peak_GBs": L2_PEAK_GBS},
        "stage_ms": {k: c.timing[k] for k in ("ms_total", "ms_sample", "ms_correlate", "ms_peaks")} if c.timing else None,
        "device": c.eng.device_name,
    }
    # the merged keys of the last timed step: the same 32 numbers at every world size
    best_host = c.best.cpu()
    out["keys_digest"] = keys_digest(best_host.numpy())
    out["keys_digest_comparable_across_n"] = not c.iq8  # (the IQ stand-in is one sequential torch stream per rank)
    if not c.grid:
        snr, lo, ca = c.gdist.unpack_keys(best_host, c.eng.kmax)
        hits = c.torch.nonzero(snr >= 25).flatten().tolist()
        out["detected_prns"] = [int(p) + 1 for p in hits]
        out["detected"] = [{"prn": int(p) + 1, "snr": round(float(snr[p]), 3), "lo_shift": int(lo[p]), "ca_shift": int(ca[p])} for p in hits]
        if not c.args.capture and c.args.data == "signals":
            out["injected_prns_all_ranks"] = sorted(set().union(*[synth_sats(CAPTURE_SEED + r, c.fs)[0] for r in range(c.world)])) if c.iq8 else c.injected
    out["extras"] = {"one_rank_process_group": c.dist is not None and c.world == 1, "soak": c.soak_leg is not None, "strong_share_at_8": c.share is not None,
                     "weak_scaling": c.weak is not None, "live_traffic": traffic_live is not None, "pk_fma_stream": c.pk_stream is not None,
                     "cpu_baseline": c.world == 1 and not c.args.no_cpu_baseline, "e2e_cli": c.world == 1 and not c.args.no_e2e,
                     "keys": "gpsacq_peak_keys_device (library, engine stream)"}
    for k, v in (("one_rank_collective", c.no_coll), ("soak", c.soak_leg), ("strong_share_at_8", c.share), ("weak_scaling", c.weak)):
        if v is not None:
            out[k] = v
    if c.iq8 and c.leg.sample_ms:
        ms = float(np.mean(c.leg.sample_ms))
        b_in, b_out = c.nblk * 80000, c.nblk * c.eng.doppler_sub * 8 * 5000 * 8
        out["ingest"] = {"kernel": "k_fwd2<iq8>", "ms": ms, "bytes_read": b_in, "bytes_written": b_out, "GBs": (b_in + b_out) / (ms * 1e-3) / 1e9,
                         "copy_ceiling_GBs": 6290.0, "frac_of_copy_ceiling": (b_in + b_out) / (ms * 1e-3) / 1e9 / 6290.0,
                         "note": "8-bit IQ read (80 000 B per block) + polyphase spectrum written (320 KB per block) over the stage's HIP-event time"}
    # ---- the oracle: baseline (N = 1) and checker (any N) -- test infrastructure, outside the timed region ----
    if c.check_parity:
        n_share = c.nblk if (c.world == 1 or c.args.parity_blocks <= 0) else min(c.nblk, c.args.parity_blocks)
        host_bits = c.d_bits[:n_share * 5120].cpu().numpy()  # (the bits path: blocks 5120 bytes apart)
        ndop, first_block = c.eng.num_doppler, c.first_run * 32
        port = port_peaks = None
        if c.world == 1:
            out["roofline"]["hbm_secondary"]["hbm_copy_measured_GBs"] = hbm_copy_gbs(c.torch, c.dev)
            port, port_peaks = cpu_baseline(c.cfg, host_bits[:min(n_share, 1024) * 5120], ndop)
            ref = cpu_baseline_reference(c.cfg, host_bits[:min(n_share, 1024) * 5120], ndop)
            out["cpu_baseline"] = ref or port
            if ref:
                out["cpu_baseline_port"] = port
            verdict_home = out["cpu_baseline"]
        else:  # the CPU baseline is an N = 1 figure; the verdict on rank 0's share is not
            verdict_home = out
        # the checker's side: the oracle's float build over rank 0's WHOLE share on every core the process may use (= the all-cores figure)
        cpu_cells = None
        try:
            allc, cpu_peaks, cpu_cells, _ = cpu_baseline_all_cores(c.cfg, host_bits, ndop, port["value"] if port else None,
                                                                    cap_s=c.args.parity_seconds, first_block=first_block)
            if port_peaks is not None:  # the same code on one thread and on many: the same peaks, bit for bit
                m = min(len(port_peaks), len(cpu_peaks))
                allc["equals_the_one_thread_run"] = bool(all(np.array_equal(port_peaks[f][:m], cpu_peaks[f][:m]) for f in ("snr", "lo_shift", "ca_shift")))
        except Exception as ex:  # (no OpenMP build, ...): the verdict falls back to the blocks one thread searched, peaks only
            allc = {"error": str(ex)[:300]}
            if port_peaks is None:
                _, port_peaks = cpu_baseline(c.cfg, host_bits, ndop, target_s=4.0, max_blocks=96)
            cpu_peaks = port_peaks
        if c.world == 1:
            out["cpu_baseline_all_cores"] = allc
        else:
            out["parity_oracle_run"] = allc
        if c.args.parity_selftest:
            c.gpu_peaks = c.gpu_peaks.copy()
            i_bad = min(7, len(c.gpu_peaks) - 1)
            c.gpu_peaks["ca_shift"][i_bad] = (c.gpu_peaks["ca_shift"][i_bad] + 1) % c.eng.num_lags
        try:
            par = parity_vs_gpu(c.cfg, c.gpu_peaks, c.gpu_cells, cpu_peaks, cpu_cells, host_bits, first_block=first_block, share_blocks=c.nblk)
        except Exception as ex:
            par = {"ok": False, "error": str(ex)[:300]}
        verdict_home["parity_vs_gpu"] = par
        # (flat copies: a consumer that keeps scalars only still sees the verdict)
        verdict_home.update({"parity_ok": par.get("ok"), "parity_blocks": par.get("blocks"), "parity_cells": par.get("cells"),
                             "parity_whole_share": par.get("whole_share"),
                             "parity_ca_equal": par.get("ca_equal"), "parity_lo_equal": par.get("lo_equal"), "parity_proven_ties": par.get("proven_ties"),
                             "parity_snr_max_rel": par.get("snr_max_rel"), "parity_pwr_max_rel": par.get("pwr_max_rel")})
        parity_failed = not par.get("ok")
    if c.world == 1 and not c.args.no_library_baseline and not c.grid and not c.iq8 and c.n_tasks >= 32:
        try:  # the same cells through the vendor FFT on this GPU
            from gpsacq import PEAK_DTYPE as _PK2
            pk = c.leg.d_peaks[(c.leg.step_no - 1) % c.leg.NBUF][:32].cpu().numpy().view(_PK2).reshape(-1)
            out["extras"]["gpu_library_baseline"] = gpu_library_baseline(c.torch, c.eng, c.dev, c.d_bits, c.stride, c.nblk, gpu_peaks=pk)
        except Exception as ex:
            out["extras"]["gpu_library_baseline"] = {"error": str(ex)[:300]}
    if c.world > 1 and not c.args.no_inproc_multi and not c.grid and not c.iq8 and not c.args.capture and c.args.data == "signals":
        out["extras"]["inproc_multi"] = inproc_multi(c.cfg, c.total_runs, CAPTURE_SEED, c.devices, out["keys_digest"])
    if c.world == 1 and not c.args.no_e2e and not c.grid and not c.iq8 and not c.args.capture and c.args.config in (1, 2):
        try:
            out["e2e_cli"] = e2e_cli(c.cfg, c.d_bits, c.nblk // 32, c.eng.num_doppler)
        except Exception as ex:
            out["e2e_cli"] = {"error": str(ex)}
    return out, parity_failed


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "inproc_multi":
        inproc_multi_child(sys.argv[2:])
    else:
        raise SystemExit("usage: bench_extras.py inproc_multi --fc .. --fs .. --max-fo .. --runs .. --seed .. --devices 0,1,..")
