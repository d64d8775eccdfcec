#!/usr/bin/env python3
"""bench.py -- (PRN, Doppler) correlation cells/s of the MI355X acquisition engine.

Contract (task statement): `python bench.py --gpus N --steps K --warmup W`; for N > 1 the driver
launches one rank per GPU through torch.distributed.run (RCCL).  Rank 0 prints ONE JSON line.

Default workload (BASELINE.json configs[1]): 32 PRN, fs = 5.456 MHz, IF = 4.092 MHz, N = 40000,
+-5 kHz -> 73 Doppler bins of fs/N, 5456 lags scanned; reference schedule (SearchTask,
c/search_offline.cpp:239-246): every 5120-byte block of the capture is searched against PRN
(block % 32) over all Doppler bins.  One step = one pass of the whole hot path (1-bit unpack + mix
+ forward FFT-40000 per block, then 73 fused multiply / IFFT-40000 / peak cells per block, then the
per-(block, PRN) peak) over a capture of `--blocks-total` blocks already resident in HBM -- 10 880
blocks = 340 runs, the size of the Nottingham capture the metric is quoted on.

Scaling: STRONG.  The capture is the same size at every N; whole runs are split over the ranks
(gpsacq.dist.shard_runs) and one RCCL all-reduce(MAX) of the 32 per-PRN best peaks (256 bytes) closes
each step.  `value` = total cells of the capture / max-over-ranks time.  A second, untimed-by-the-driver
leg with `--weak-blocks` blocks on EVERY rank (per-GPU work fixed) is reported under "weak_scaling".

Other configurations (the default line stays the one the driver records):
  --config 2   BASELINE configs[2]: fs 8.184 MHz / IF 2.046 MHz (gps_sig_gen.m's rates), 49 bins, 8184 lags
  --config 3   BASELINE configs[3]: rtl-sdr path, fs 2.8 MHz, +-100 kHz, 5 non-coherent sums
  --config 4   BASELINE configs[4]: one capture x 32 PRN x +-100 kHz fine grid, Doppler slabs over the ranks
  --input iq8  the capture is an 8-bit IQ stream (rtl-sdr uint8, README.md:83-115) resident in HBM and searched directly:
               mean removal, mixer, sign and bit transpose are fused into the forward transform (no 1-bit intermediate);
               the line gains "ingest" = bytes moved by that stage / its time, against the 6.29 TB/s copy ceiling
  --capture F  search the 1-bit capture file F (e.g. gps.samples.1bit.I.fs5456.if4092.bin) instead of
               synthetic data: whole runs of the file, same schedule, and the SearchTask report's hit list

At N = 1 the process still joins a ONE-RANK `nccl` (= RCCL) process group by default, so the collective code of the N > 1
runs -- the per-step all-reduce(MAX) of the packed keys on torch's stream, `dist.barrier()` in the fences, the device-side
all-reduce of the elapsed time -- executes on every run of this file (`dist_backend: "nccl"`, `rccl_ranks_seen: 1`);
`--no-dist` skips it, `--force-dist` makes a failure to create the group fatal instead of falling back.

The line verifies itself and is comparable across boxes (round 5):
  cpu_baseline.parity_vs_gpu  the peaks of the LAST TIMED STEP against the oracle (test infrastructure, the checker) at the configuration
        measured: every block the cpu_baseline leg pushes through the oracle's float build must carry the same ca_shift / lo_shift
        (or a tie proven in the double-precision oracle) and an SNR within 1e-4, and three seeded random full rows of cells of the
        whole capture must agree with liboracle_f64 to 2e-5 (c/search_offline.cpp:190-198,248).  Flat copies (parity_ok,
        parity_blocks, ...) sit beside it.  A disagreement makes the run EXIT 3 after the line is printed.
  roofline.sclk_mhz / power_w   the clock over the K timed steps from the GPU's own cycle counters (s_memtime of all 8 XCDs, stamped on the
        engine's stream before and after; mean over the XCDs -- they run up to 5 % apart under the power cap: sclk_mhz_xcd_min / _max),
        with the sysfs readings (pp_dpm_sclk -- XCD 0's clock --, hwmon power1; every 100 ms during the steps) beside it
  roofline.cycles_per_cell_per_cu = kernel_ms x sclk x CUs / cells, roofline.frac_at_clock = achieved / (157.3 TFLOP/s x sclk / 2.4 GHz)
  roofline.pk_fma_stream_TF     a pure v_pk_fma_f32 stream at k_corr's residency for 3 s in the untimed part (gnss-gps-sdr_amd/bin/
        pk_fma_stream): what the fp32 vector pipe sustains on this box, next to the datasheet's 157.3
  extras / one_rank_collective  which optional legs ran beside the timed steps, and the N = 1 step with and without its process group
The per-step merge keys are made by the library (gpsacq_peak_keys_device, one launch on the engine's stream): a step is one search,
the keys, one all-reduce.

After the K timed steps a `soak` leg repeats the same step until >= 6 s of GPU time have passed (reported separately; `steps`
and `ms_per_step` are untouched) with one sclk / package-power sample taken in mid-leg, so that a coarse SMI sampler beside
the run sees the GPU busy at the rate the line claims.

`--gpus N` with N > 1 and no WORLD_SIZE in the environment re-launches itself under torch.distributed.run with N ranks
(so a plain `python bench.py --gpus 8` cannot silently measure one GPU); fewer than N visible devices is an error.
"""
import argparse
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
FP32_VALU_PEAK_TF = 157.3        # MI355X_MICROARCH.md: peak FP32 vector (= FP32 MFMA) rate, dense
FP32_PEAK_CLOCK_MHZ = 2400.0     # the clock that figure assumes: 256 CUs x 128 lanes x 2 flop x 2.4 GHz
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
L2_PEAK_GBS = 34500.0            # MI355X_MICROARCH.md: aggregate L2 bandwidth
ALG_BYTES_PER_CELL = 32 * N_FFT  # read signal + code spectra, write + read one IFFT intermediate

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


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


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


def smi_sample(dev_index=0):
    """One sclk / package power / junction temperature reading of the GPU (rocm-smi; best effort -- None fields when the tool
    or a field is missing)."""
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


def soak(leg, cells_job, min_gpu_s=6.0, max_steps=2000):
    """The timed step again until >= min_gpu_s of GPU time: steps are enqueued in slices (the engine's stream is the clock:
    the wall time of a slice that ends in a synchronize), an SMI reading is taken from a thread in mid-leg."""
    import threading
    readings = []
    th = None
    done_s, steps, slices = 0.0, 0, []
    torch = leg.torch
    stamps = torch.zeros((2, STAMP_SLOTS), dtype=torch.int64, device=leg.dev)  # every CU's cycle counter around the whole leg
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
            "sclk_mhz": chip_mhz, "sclk_mhz_per_xcd": per_xcd,  # over the whole leg, from the CUs' own cycle counters (mean of the XCDs' medians)
            "smi_mid_leg": readings[0] if readings else None,
            "note": "same step as the timed region, repeated after it; reported separately, `steps`/`ms_per_step`/`value` are the K timed steps"}


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
    """One reading of the shader clock (the level pp_dpm_sclk marks with '*' -- what rocm-smi --showclocks prints) and the package
    power (hwmon, microwatts) straight from sysfs: a file read, cheap enough to repeat every 100 ms beside a timed leg."""
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


STAMP_SLOTS = 512  # gpsacq.h GPSACQ_STAMP_SLOTS


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


def pk_fma_stream(seconds=3.0):
    """gnss-gps-sdr_amd/bin/pk_fma_stream (tools/ubench/pk_fma_stream.hip): the rate of a pure v_pk_fma_f32 stream at k_corr's
    residency for `seconds` -- the practical ceiling of the fp32 vector pipe on this box under load.  Returns its JSON or {"error"}."""
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


PARITY_SNR_REL = 1e-4   # BASELINE.json north_star: correlator magnitudes within 1e-4 relative
PARITY_PWR_REL = 2e-5   # what the GPU suite asserts per cell (tests/test_gpu_parity.py REL)
PARITY_TIE_REL = 1e-5   # two candidates closer than this in the double-precision oracle are a float-rounding tie


def compare_peaks(cfg, gpu_peaks, cpu_peaks, host_bits):
    """Part (1) of parity_vs_gpu, host only: the GPU's peaks against the oracle's for the same blocks of the same capture (reference
    schedule: block b against PRN b % 32).  Returns (dict, oracle_f64, lag_powers) -- the double-precision oracle and its per-lag
    power probe are reused by part (2)."""
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
    ties, unproven = 0, []
    for b in np.nonzero(~(ca_eq & lo_eq))[0]:
        blk = np.frombuffer(host_bits, dtype=np.uint8)[b * 5120:(b + 1) * 5120]
        snr2 = []
        for pk in (g[b], o[b]):
            if not (-orc.dmax <= int(pk["lo_shift"]) <= orc.dmax and 0 <= int(pk["ca_shift"]) < S):
                snr2.append(float("nan"))  # a result outside the search grid is never a tie
                continue
            pw = lag_powers(blk, int(b) % 32, pk["lo_shift"])
            snr2.append(float(pw[pk["ca_shift"]]) / (float(pw.sum(dtype=np.float64)) / S))
        if abs(snr2[0] - snr2[1]) <= PARITY_TIE_REL * snr2[1]:
            ties += 1
        else:
            unproven.append(int(b))
    with np.errstate(divide="ignore", invalid="ignore"):
        snr_rel = np.abs(g["snr"].astype(np.float64) / o["snr"].astype(np.float64) - 1.0)
    snr_max_rel = float(np.nanmax(snr_rel)) if n else 0.0
    return ({"blocks": int(n), "ca_equal": int(ca_eq.sum()), "lo_equal": int(lo_eq.sum()), "proven_ties": ties, "unproven_mismatches": unproven[:8],
             "n_unproven": len(unproven), "snr_max_rel": snr_max_rel}, orc, lag_powers)


def parity_vs_gpu(cfg, eng, torch, d_bits, nblk, stride, gpu_peaks, cpu_peaks, host_bits, seed=5):
    """The timed step's own results against the oracle (test infrastructure; c/search_offline.cpp:190-198,248), AFTER the timed
    region: (1) the GPU's peak of every block the cpu_baseline leg pushed through the oracle's float build -- same capture, same
    blocks, reference schedule block -> PRN block % 32 -- must carry the same ca_shift and lo_shift and an SNR within 1e-4; a
    different (lo, ca) is accepted only as a PROVEN tie: in the double-precision oracle the two candidates' SNRs agree to 1e-5.
    (2) three seeded random (block, PRN) rows of the WHOLE capture, all Doppler bins, cell by cell against liboracle_f64:
    max_pwr / tot_pwr to 2e-5, the lag identical or a proven tie."""
    from oracle_lib import CELL_DTYPE
    part1, orc, lag_powers = compare_peaks(cfg, gpu_peaks, cpu_peaks, host_bits)
    dmax = orc.dmax
    # (2) full rows of cells: GPU cells of three tasks through the same C ABI, device-resident capture
    rng = np.random.default_rng(seed)
    rows = sorted(int(b) for b in rng.choice(nblk, size=min(3, nblk), replace=False))
    dev = d_bits.device
    tasks = torch.tensor([[b, b % 32] for b in rows], dtype=torch.int32, device=dev)
    d_cells = torch.zeros((len(rows), eng.num_doppler, 4), dtype=torch.int32, device=dev)
    d_pk = torch.zeros((len(rows), 4), dtype=torch.int32, device=dev)
    eng.search_device(d_bits.data_ptr(), nblk, d_pk.data_ptr(), stride=stride, d_tasks_ptr=tasks.data_ptr(), n_tasks=len(rows),
                      d_cells_ptr=d_cells.data_ptr(), sync=True)
    gc = d_cells.cpu().numpy().view(CELL_DTYPE).reshape(len(rows), eng.num_doppler)
    pwr_max_rel, lag_ties, lag_bad, cells = 0.0, 0, [], 0
    for r, b in enumerate(rows):
        blk = d_bits[b * stride:b * stride + 5120].cpu().numpy()
        oc, _ = orc.search_block(blk, b % 32)
        cells += oc.size
        pwr_max_rel = max(pwr_max_rel, float(np.max(np.abs(gc[r]["max_pwr"] / oc["max_pwr"] - 1.0))), float(np.max(np.abs(gc[r]["tot_pwr"] / oc["tot_pwr"] - 1.0))))
        for d in np.nonzero(gc[r]["max_i"] != oc["max_i"])[0]:
            pw = lag_powers(blk, b % 32, int(d) - dmax)
            a, c = float(pw[gc[r]["max_i"][d]]), float(pw[oc["max_i"][d]])
            if abs(a - c) <= PARITY_TIE_REL * c:
                lag_ties += 1
            else:
                lag_bad.append([int(b), int(d) - dmax])
    ok = part1["n_unproven"] == 0 and part1["snr_max_rel"] <= PARITY_SNR_REL and pwr_max_rel <= PARITY_PWR_REL and not lag_bad
    return {"ok": bool(ok), "blocks": part1["blocks"], "ca_equal": part1["ca_equal"], "lo_equal": part1["lo_equal"], "proven_ties": part1["proven_ties"],
            "unproven_mismatches": part1["unproven_mismatches"], "snr_max_rel": part1["snr_max_rel"], "cells": int(cells), "cell_rows_block_prn": [[b, b % 32] for b in rows],
            "pwr_max_rel": pwr_max_rel, "cell_lag_ties": lag_ties, "cell_lag_mismatches": lag_bad[:8],
            "tolerances": {"snr_rel": PARITY_SNR_REL, "pwr_rel": PARITY_PWR_REL, "tie_rel": PARITY_TIE_REL},
            "what": "GPU peaks of the LAST TIMED STEP vs the oracle's float build on the blocks the cpu_baseline leg searched (ca_shift / lo_shift equal "
                    "or a tie proven in the double-precision oracle, SNR to 1e-4) + 3 random full rows of cells vs liboracle_f64 (2e-5)"}


def live_traffic(argv_tail, timeout_s=150):
    """HBM bytes of one k_corr launch measured IN THIS RUN: two short child runs of this file under `rocprofv3 --pmc FETCH_SIZE`
    and `--pmc WRITE_SIZE` (separate passes, as /opt/skills/guides/MI355X_MICROARCH.md prescribes; no trace domains beside
    --pmc), same workload, 2 timed steps.  FETCH_SIZE is in KiB and counts half the bytes of wide coalesced reads on gfx950
    (x 2, the guide's correction); WRITE_SIZE KiB is uncalibrated.  Returns {"read_bytes", "write_bytes", "launches"} or
    {"error": ...}; never raises."""
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
            cmd = ["rocprofv3", "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "p", "--", sys.executable, os.path.abspath(__file__),
                   "--steps", "2", "--warmup", "1", "--weak-blocks", "0", "--no-cpu-baseline", "--no-e2e", "--soak-seconds", "0", "--no-dist",
                   "--no-live-traffic"] + argv_tail
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


def cpu_baseline(cfg, bits, ndop, target_s=12.0):
    """The oracle's float build (own mixed-radix FFT; `port`) timed single-threaded on a bounded sample of the
    same capture.  The reference binary itself cannot run on the GPU box (it needs FFTW; oracle/_ref/README)."""
    from oracle_lib import Oracle
    orc = Oracle(cfg["fc"], cfg["fs"], cfg["max_fo"], kind="f32")
    t0 = time.perf_counter()
    cells, _ = orc.bench_blocks(bits[:2 * 5120], 2)
    dt = time.perf_counter() - t0
    nblk = int(max(2, min(len(bits) // 5120, target_s / (dt / 2))))
    t0 = time.perf_counter()
    cells, peaks = orc.bench_blocks(bits[:nblk * 5120], nblk)
    dt = time.perf_counter() - t0
    cpu_baseline.last_peaks = peaks  # the checker's answers for exactly these blocks: parity_vs_gpu compares the timed step's with them
    return {"value": cells / dt, "unit": "cells/s", "cores": 1, "kind": "port",
            "sample": f"{nblk} blocks x {ndop} bins = {cells} cells of the same capture, oracle f32 build (own FFT, -O3), "
                      f"{dt:.1f} s on {os.cpu_count()} core host ({cpu_model()}), 1 thread",
            "context": "the reference itself needs FFTW3f, absent from this image and from the GPU box (oracle/_ref unbuildable); the survey "
                       "timed the reference's sources against an MKL FFT stand-in at 3.2-5.7 k cells/s per core of a 2.1 GHz Xeon (BASELINE.md "
                       "section 2; SURVEY.md section 8d) -- the same range as this port, so the port is a fair stand-in for the reference's CPU rate"}


def cpu_baseline_reference(cfg, bits, ndop, target_s=15.0):
    """The reference binary itself (oracle/_ref/gps_test_ref, built by `make -C oracle ref` where a real FFTW3 exists;
    travels to the GPU box with the snapshot) timed on whole runs of the same capture.  None if it was never built."""
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


def kernel_source_sha():
    """sha256 over the sources of the timed kernels: profiles/traffic.json carries the value of the tree it was profiled
    on (tools/summarize_prof.py), so counters from another kernel binary are flagged instead of shipped silently."""
    import hashlib
    h = hashlib.sha256()
    for name in ("acq_kernels.hip", "acq_phases.hpp", "acq_math.hpp"):
        with open(os.path.join(ROOT, "gnss-gps-sdr_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def cpu_baseline_all_cores(cfg, bits, ndop, one_thread_rate, target_s=8.0):
    """The same port on every core the process may use: an OpenMP loop inside liboracle_f32.so (oracle_bench_omp: one
    oracle instance per thread, SearchInit untimed, blocks dealt round-robin for `target_s` seconds)."""
    import ctypes
    from oracle_lib import lib
    L = lib("f32")
    L.oracle_bench_omp.restype = ctypes.c_long
    L.oracle_bench_omp.argtypes = [ctypes.c_double] * 3 + [ctypes.c_void_p, ctypes.c_long, ctypes.c_long, ctypes.c_int, ctypes.c_double,
                                                            ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)]
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
    el, used = ctypes.c_double(), ctypes.c_int()
    nthreads = ncpu if quota is None else max(1, min(ncpu, int(math.ceil(quota))))  # more threads than the quota only add switching
    cells = L.oracle_bench_omp(cfg["fc"], cfg["fs"], cfg["max_fo"], buf.ctypes.data, nblk, 5120, nthreads, target_s, ctypes.byref(el), ctypes.byref(used))
    rate = cells / el.value
    return {"value": rate, "unit": "cells/s", "cores": used.value, "kind": "port",
            "sched_getaffinity_cores": ncpu, "os_cpu_count": os.cpu_count(), "cgroup_cpu_quota_cores": quota,
            "speedup_over_1_thread": rate / one_thread_rate if one_thread_rate else None,
            "sample": f"OpenMP, {used.value} threads (sched_getaffinity: {ncpu} cores, cgroup cpu.max quota: {quota} cores), blocks of the same {nblk}-block sample dealt "
                      f"round-robin x {ndop} bins for {el.value:.1f} s = {cells} cells"}


def e2e_cli(cfg, d_bits, n_runs, ndop, reps=5):
    """The drop-in a user runs: wall clock of gnss-gps-sdr_amd/bin/gps_test on a capture FILE of the bench's size (written
    from the resident synthetic capture), process start to exit, with the front end's own split (GPSACQ_TRACE)."""
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


class Leg:
    """One timed workload: `n_tasks` tasks over `nblk` resident blocks on this rank."""

    def __init__(self, torch, gpsacq, gdist, eng, dev, dist, backend, nblk, n_tasks, d_bits, d_tasks, stride, grid, n_keys=32, iq=None):
        self.torch, self.gdist, self.eng, self.dev, self.dist, self.backend = torch, gdist, eng, dev, dist, backend
        self.iq = iq  # gpsacq.Iq8Input: d_bits then holds interleaved 8-bit I,Q bytes, `stride` bytes per block
        self.nblk, self.n_tasks, self.d_bits, self.d_tasks, self.stride, self.grid, self.n_keys = nblk, n_tasks, d_bits, d_tasks, stride, grid, n_keys
        # The search runs on the engine's own HIP stream; the key packing and the collective run on torch's
        # stream, ordered after it by an event, so step i's reduction / all-reduce overlaps step i+1's search
        # (two peak buffers; the engine stream waits for a buffer's previous reader).
        self.d_peaks = [torch.zeros((max(n_tasks, 1), 4), dtype=torch.int32, device=dev) for _ in range(2)]
        # the merge keys (gpsacq_peak_keys_device: one launch on the engine's stream right behind the search): 32 per-PRN best keys
        # on the block schedule, one per task on the grid; a rank without work keeps zeros (neutral for MAX)
        self.d_keys = [torch.zeros(n_keys, dtype=torch.int64, device=dev) for _ in range(2)]
        self.sampler = None  # ClockSampler: sclk / power readings during the timed steps of run()
        self.eng_stream = torch.cuda.ExternalStream(eng.stream_ptr, device=dev)
        self.reader_done = [None, None]
        self.step_no = 0

    def step(self):
        torch, eng = self.torch, self.eng
        slot = self.step_no & 1
        self.step_no += 1
        buf, best = self.d_peaks[slot], self.d_keys[slot]
        if self.n_tasks > 0:
            if self.reader_done[slot] is not None:
                self.eng_stream.wait_event(self.reader_done[slot])
            if self.iq is not None:
                eng.search_iq8_device(self.d_bits.data_ptr(), self.iq, self.nblk, buf.data_ptr(), stride=self.stride, sync=False)
            else:
                eng.search_device(self.d_bits.data_ptr(), self.nblk, buf.data_ptr(), stride=self.stride,
                                  d_tasks_ptr=self.d_tasks.data_ptr() if self.d_tasks is not None else None,
                                  n_tasks=self.n_tasks, sync=False)
            # best peak per PRN (block schedule) / per (block, PRN) (grid), packed so that integer MAX reproduces the reference's
            # ordering (higher SNR; ties -> lower Doppler bin, :198): made by the library, one launch on the engine's stream
            eng.peak_keys_device(buf.data_ptr(), self.n_tasks, best.data_ptr(), per_prn=not self.grid, sync=False)
            searched = torch.cuda.Event()
            searched.record(self.eng_stream)
            torch.cuda.current_stream().wait_event(searched)
        # (a rank without work -- more ranks than runs / grid points -- contributes keys of 0, neutral for MAX)
        if self.dist is not None:
            if self.backend == "nccl":
                self.dist.all_reduce(best, op=self.dist.ReduceOp.MAX)  # RCCL over xGMI, 256 bytes
            else:
                b = best.cpu()
                self.dist.all_reduce(b, op=self.dist.ReduceOp.MAX)
                best.copy_(b)
        self.reader_done[slot] = torch.cuda.Event()
        self.reader_done[slot].record(torch.cuda.current_stream())
        return best

    def fence(self):
        self.eng.synchronize()
        self.torch.cuda.synchronize()
        if self.dist is not None:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def run(self, steps, warmup):
        """W untimed + K timed steps bracketed by barrier + synchronize; returns (max-over-ranks seconds, mean
        correlate-kernel ms on this rank, last best keys)."""
        torch = self.torch
        best = None
        for _ in range(warmup):
            best = self.step()
        self.fence()
        corr_ms, self.sample_ms = [], []
        stamps = ev = None
        if self.sampler is not None:
            self.sampler.start()
            if self.n_tasks > 0:  # shader-cycle stamps + HIP events on the engine's stream around the timed steps: the clock the GPU itself counted
                stamps = torch.zeros((2, STAMP_SLOTS), dtype=torch.int64, device=self.dev)  # [before / after][xcc << 6 | se << 4 | cu]
                ev = [torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)]
                self.eng.cycle_stamp_device(stamps[0].data_ptr())
                ev[0].record(self.eng_stream)
        t0 = time.perf_counter()
        for i in range(steps):
            best = self.step()
            if i > 0 and self.n_tasks > 0:  # the previous search's times: waits for that search only
                tm = self.eng.last_timing(1)
                corr_ms.append(tm["ms_correlate"])
                self.sample_ms.append(tm["ms_sample"])
        if self.n_tasks > 0:
            tm = self.eng.last_timing(0)
            corr_ms.append(tm["ms_correlate"])
            self.sample_ms.append(tm["ms_sample"])
        if stamps is not None:
            self.eng.cycle_stamp_device(stamps[1].data_ptr())
            ev[1].record(self.eng_stream)
        self.fence()
        elapsed = time.perf_counter() - t0
        self.memtime_mhz = self.memtime_per_xcd = self.memtime_cus = None
        if stamps is not None:
            ms = ev[0].elapsed_time(ev[1])
            self.memtime_mhz, self.memtime_per_xcd, self.memtime_cus = cu_clocks(stamps.cpu().numpy(), ms)
        if self.sampler is not None:
            self.sampler.stop()
        t = torch.tensor([elapsed], dtype=torch.float64, device=self.dev if self.backend == "nccl" else "cpu")
        if self.dist is not None:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item()), (float(np.mean(corr_ms)) if corr_ms else 0.0), (best.clone() if best is not None else None)


_REAL_STDOUT = None


def claim_stdout():
    """From here on file descriptor 1 is stderr, and the JSON line is written to the saved descriptor of the real stdout:
    libraries that print to the C-level stdout (RCCL writes a five-line version banner there when its communicator comes up,
    flushed at exit, i.e. AFTER the result) can no longer add lines to what the driver parses."""
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.dup(1)
        os.dup2(2, 1)


def emit(line):
    data = (line + "\n").encode()
    if _REAL_STDOUT is None:
        sys.stdout.write(line + "\n")
        sys.stdout.flush()
    else:
        os.write(_REAL_STDOUT, data)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # 100 steps x 45 ms: a GPU leg long enough for a 5-second SMI sampler to see it (the two CPU baselines take 20 s beside it)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, choices=[1, 2, 3, 4], default=1)
    ap.add_argument("--blocks-total", type=int, default=10880,
                    help="blocks of the whole capture (all ranks together; whole runs of 32): 10880 = the Nottingham capture")
    ap.add_argument("--weak-blocks", type=int, default=4096, help="blocks per rank of the weak-scaling leg (0: skip it)")
    ap.add_argument("--grid-blocks", type=int, default=4, help="--config 3/4: capture positions searched against all 32 PRNs")
    ap.add_argument("--doppler-step", type=float, default=0.0,
                    help="--config 3/4: requested Doppler step in Hz (0: the FFT bin fs/N); the engine takes the finest grid "
                         "it has that is not coarser (sub-bin phase ramps) or the coarsest not finer (bin stride)")
    ap.add_argument("--capture", default=None, help="1-bit capture file to search instead of synthetic data (config 1/2 schedules)")
    ap.add_argument("--input", choices=["bits", "iq8"], default="bits",
                    help="iq8: an 8-bit IQ capture (uint8, offset 128) converted inside the forward transform (configs 1-3 schedules)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-dist", action="store_true", help="N = 1: do not create the one-rank nccl process group")
    ap.add_argument("--force-dist", action="store_true", help="N = 1: the one-rank nccl group must come up (no fallback)")
    ap.add_argument("--soak-seconds", type=float, default=6.0, help="GPU time of the soak leg after the timed steps (0: skip)")
    ap.add_argument("--live-traffic", action="store_true", help="(kept for old command lines: live traffic is on unless --no-live-traffic)")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="do not measure roofline.traffic in this run (two short rocprofv3 --pmc child runs); use profiles/traffic.json")
    ap.add_argument("--parity-selftest", action="store_true",
                    help="corrupt one GPU peak (ca_shift + 1) before cpu_baseline.parity_vs_gpu compares: the run must then exit 3 (tests)")
    ap.add_argument("--pk-fma-seconds", type=float, default=3.0,
                    help="N = 1, untimed part: seconds of the pure v_pk_fma_f32 stream that gives roofline.pk_fma_stream_TF (0: skip)")
    ap.add_argument("--no-e2e", action="store_true", help="skip the gps_test end-to-end leg")
    ap.add_argument("--spawn-check", action="store_true",
                    help="launcher check without a GPU: every rank joins a gloo group, rank 0 prints the ranks it saw, all exit")
    ap.add_argument("--data", choices=["signals", "noise"], default="signals",
                    help="signals (default): capture generated on the device, white noise + 8 PRNs at seeded Doppler / code "
                         "phase (SURVEY section 8d throughput set); noise: uniform random bits")
    args = ap.parse_args()

    backend = os.environ.get("GPSACQ_DIST_BACKEND", "nccl")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # launched plainly: become N ranks (never report n_gpus = 1 for --gpus N)
        import socket
        import subprocess
        import torch
        ndev = torch.cuda.device_count()
        if backend == "nccl" and ndev < args.gpus:
            print(f"bench.py: --gpus {args.gpus} but {ndev} device(s) visible", file=sys.stderr)
            raise SystemExit(2)
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))

    claim_stdout()
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.spawn_check:
        import torch.distributed as dist
        seen = 1
        if world > 1:
            dist.init_process_group(backend="gloo")
            t = torch.ones(1, dtype=torch.int64)
            dist.all_reduce(t)
            seen = int(t.item())
            dist.destroy_process_group()
        if rank == 0:
            emit(json.dumps({"spawn_check": True, "n_gpus": args.gpus, "rccl_ranks_seen": seen, "world_size_env": world}))
        return
    import gpsacq
    from gpsacq import dist as gdist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    # GPSACQ_DIST_BACKEND=gloo lets several ranks share one GPU to exercise the N > 1 code path on a 1-GPU box
    # (collectives then run on CPU copies); the driver's runs use nccl (= RCCL).
    dev_index = local_rank % torch.cuda.device_count() if backend == "gloo" else local_rank
    if dev_index >= torch.cuda.device_count():  # more ranks than GPUs on the RCCL backend: never share a device silently
        raise SystemExit(f"rank {rank}: device {dev_index} not visible ({torch.cuda.device_count()} device(s), --gpus {args.gpus})")
    torch.cuda.set_device(dev_index)
    dist = None
    dist_note = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend=backend)
    elif not args.no_dist and backend == "nccl":
        # N = 1: a one-rank RCCL group, so that every collective line of the N > 1 runs executes here too
        import socket
        import torch.distributed as dist
        try:
            own_port = "MASTER_ADDR" not in os.environ or "MASTER_PORT" not in os.environ
            for attempt in range(5):
                if own_port:
                    with socket.socket() as sk:
                        sk.bind(("127.0.0.1", 0))
                        os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
                    os.environ["MASTER_ADDR"] = "127.0.0.1"
                try:
                    dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", dev_index))
                    break
                except Exception as ex:  # the port probed above can be taken again before the store binds it: pick another
                    if not (own_port and attempt < 4 and ("EADDRINUSE" in str(ex) or "address already in use" in str(ex))):
                        raise
            probe = torch.ones(1, dtype=torch.int64, device=torch.device("cuda", dev_index))
            dist.all_reduce(probe, op=dist.ReduceOp.MAX)  # the communicator is created lazily: bring it up before anything is timed
            torch.cuda.synchronize()
        except Exception as ex:
            if args.force_dist:
                raise
            dist_note = f"one-rank nccl group unavailable ({str(ex)[:160]}): ran without a process group"
            try:
                if dist.is_initialized():
                    dist.destroy_process_group()
            except Exception:
                pass
            dist = None

    ranks_seen = dist.get_world_size() if dist is not None else 1
    if ranks_seen != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the process group has {ranks_seen} rank(s)")

    cfg = CONFIGS[args.config]
    iq8 = args.input == "iq8"
    grid = args.config in (3, 4) and not iq8  # an IQ capture is searched block by block (reference schedule) at the config's rates
    if iq8 and args.capture:
        raise SystemExit("--input iq8 works on the synthetic capture")
    eng = gpsacq.Engine(cfg["fc"], cfg["fs"], cfg["max_fo"], device=dev_index)
    dev = torch.device("cuda", dev_index)
    fs = cfg["fs"]
    extra = {}
    stride = 5120

    if grid:
        # every rank holds the same short capture and searches all 32 PRNs over ITS slab of the Doppler grid
        n_acc = 5 if args.config == 3 else 1
        if args.doppler_step > 0:
            eng.set_doppler_step(args.doppler_step)
        if n_acc > 1:
            stride = eng.aligned_stride()
            eng.set_noncoherent(n_acc, 1)
        npos = args.grid_blocks
        nblk = npos + n_acc - 1
        data_seed = 77
        first, nbins = gdist.shard_doppler_grid(eng.num_doppler_total, eng.first_doppler_total, rank, world)
        total_bins = eng.num_doppler_total
        if nbins > 0:
            eng.set_doppler_window(first, nbins)
        tasks = np.array([(b, sv) for b in range(npos) for sv in range(32)], dtype=np.int32)
        d_tasks = torch.from_numpy(tasks).to(dev)
        n_tasks = tasks.shape[0] if nbins > 0 else 0
        cells_rank = tasks.shape[0] * nbins * n_acc
        cells_job = tasks.shape[0] * total_bins * n_acc
        workload = (f"{cfg['name']}: {npos} capture positions x 32 PRN x {total_bins} Doppler points "
                    f"(+-{cfg['max_fo'] / 1e3:.0f} kHz, step {eng.doppler_step_hz:.2f} Hz), N=40000, {eng.num_lags} lags"
                    + (f", {n_acc} non-coherent sums (blocks {stride} bytes apart)" if n_acc > 1 else ""))
        parallelism = f"Doppler slabs over {world} GPU(s), per-(position, PRN) peak all-reduce(MAX)"
        weak_blocks = 0
    else:
        if args.capture:
            size = os.path.getsize(args.capture)
            total_runs = size // (32 * 5120)  # SearchTask stops at the first short read (:241-244)
        else:
            total_runs = max(world, args.blocks_total // 32)
        first_run, n_runs = gdist.shard_runs(total_runs, rank, world)
        nblk = n_runs * 32
        data_seed = 1000 + rank
        d_tasks, n_tasks = None, nblk
        cells_rank = nblk * eng.num_doppler
        cells_job = total_runs * 32 * eng.num_doppler
        workload = (f"{cfg['name']}: 32 PRN x {eng.num_doppler} Doppler bins (+-{cfg['max_fo'] / 1e3:.0f} kHz, fs/N = {fs / N_FFT:.1f} Hz), N=40000, "
                    f"{eng.num_lags} lags, reference schedule block->PRN (block % 32); capture of {total_runs * 32} blocks "
                    f"({total_runs} runs)" + (f" = file {os.path.basename(args.capture)}" if args.capture else ""))
        parallelism = f"whole runs split over {world} GPU(s) (strong scaling), per-PRN peak all-reduce(MAX) of 256 bytes"
        weak_blocks = 0 if args.capture else args.weak_blocks

    # input resident in HBM before anything is timed
    injected, sats = synth_sats(data_seed, fs)

    def make_iq_capture(n_blocks, seed):
        """rtl-sdr style capture on the device (torch): uint8 offset-128 interleaved I,Q at BASEBAND -- complex noise of
        sigma 30 + the seeded PRNs at their Doppler + a DC offset; the engine mixes it up to the config's IF
        (proc_rtl_bin_for_gps.m:31-47) inside the forward transform.  Returns (bytes, (mean_i, mean_q))."""
        g = torch.Generator(device=dev)
        g.manual_seed(seed)
        out = torch.empty(n_blocks * 81920, dtype=torch.uint8, device=dev)
        chunk = 64
        tables = list(synth_sats(seed, fs)[1])
        chip_tab = {}
        for prn, *_ in tables:  # the 1023 chips of each PRN (G1 = x^10+x^3+1, G2 = x^10+x^9+x^8+x^6+x^3+x^2+1, c/cacode.h:9-35)
            g1, g2 = [1] * 10, [1] * 10
            t1, t2 = PRN_TAPS[prn - 1]
            c = []
            for _ in range(1023):
                c.append(g1[9] ^ g2[t1 - 1] ^ g2[t2 - 1])
                f1 = g1[2] ^ g1[9]
                f2 = g2[1] ^ g2[2] ^ g2[5] ^ g2[7] ^ g2[8] ^ g2[9]
                g1 = [f1] + g1[:9]
                g2 = [f2] + g2[:9]
            chip_tab[prn] = torch.tensor([1.0 - 2.0 * v for v in c], dtype=torch.float64, device=dev)
        for b0 in range(0, n_blocks, chunk):
            nb = min(chunk, n_blocks - b0)
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

    def make_capture(n_blocks, seed, blk_stride):
        if n_blocks == 0:
            return torch.zeros(5120, dtype=torch.uint8, device=dev)
        nbytes = (n_blocks - 1) * blk_stride + 5120
        if args.data == "signals":
            d = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            eng.generate_device(d.data_ptr(), nbytes, synth_sats(seed, fs)[1], noise_sigma=1.0, seed=seed)
            return d
        return torch.from_numpy(np.random.default_rng(seed).integers(0, 256, size=nbytes, dtype=np.uint8)).to(dev)

    iq_in = None
    if iq8:
        stride = 81920
        d_bits, iq_mean = make_iq_capture(max(nblk, 1), data_seed)
        iq_in = eng.iq8_input(signed=False, remove_dc=True, mean=iq_mean, mix_hz=cfg["fc"], fs=fs, first_sample=0, total_samples=max(nblk, 1) * 40960)
    elif args.capture and not grid:
        with open(args.capture, "rb") as f:
            f.seek(first_run * 32 * 5120)
            host = np.frombuffer(f.read(nblk * 5120), dtype=np.uint8)
        d_bits = torch.from_numpy(host.copy()).to(dev) if nblk else torch.zeros(5120, dtype=torch.uint8, device=dev)
    else:
        d_bits = make_capture(nblk, data_seed, stride)

    # every rank's share of the work, gathered before anything is timed (what SCALE lines are checked against)
    blocks_per_rank = [nblk]
    if dist is not None:
        t = torch.zeros(world, dtype=torch.int64, device=dev if backend == "nccl" else "cpu")
        t[rank] = n_tasks if grid else nblk
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        blocks_per_rank = [int(v) for v in t.tolist()]

    leg = Leg(torch, gpsacq, gdist, eng, dev, dist, backend, nblk, n_tasks, d_bits, d_tasks, stride, grid,
              n_keys=(tasks.shape[0] if grid else 32), iq=iq_in)
    if rank == 0:
        leg.sampler = ClockSampler(torch, dev_index)  # sclk / power during the K timed steps (roofline.sclk_mhz, .power_w)
    elapsed, kern_ms, best = leg.run(args.steps, args.warmup)
    timing = eng.last_timing() if n_tasks > 0 else None
    clock = leg.sampler.stats() if leg.sampler is not None else None
    if clock is not None:
        # the clock of the roofline: the GPU's own cycle counts over the timed steps (s_memtime stamps of all 8 XCDs / HIP events on the
        # engine's stream), averaged over the XCDs -- under the power cap they run up to 5 % apart, and the sysfs reading follows XCD 0
        # alone (profiles/r05_experiments/e_xcd_clocks.log).  The sysfs readings stay beside it and take over only if the stamps are unusable
        clock["sclk_mhz_sysfs"] = clock["sclk_mhz"]
        mt = getattr(leg, "memtime_mhz", None)
        clock["sclk_mhz_cycle_counter"] = mt
        clock["sclk_mhz_per_xcd"] = getattr(leg, "memtime_per_xcd", None)
        clock["cycle_counter_cus"] = getattr(leg, "memtime_cus", None)
        good = [m for m in (clock["sclk_mhz_per_xcd"] or []) if m]
        clock["sclk_mhz_xcd_min_max"] = [min(good), max(good)] if good else None
        if mt:
            clock["sclk_mhz"] = mt
            clock["source"] = "shader-cycle counters of every CU (s_memtime stamps around the timed steps / HIP-event time between them): mean of the per-XCD medians; sysfs beside it: " + clock["source"]
    leg.sampler = None
    # the peaks of the LAST TIMED STEP, kept for the parity verdict (cpu_baseline.parity_vs_gpu) before any other leg runs
    gpu_peaks = None
    if rank == 0 and world == 1 and not grid and not iq8 and n_tasks > 0 and not args.no_cpu_baseline:
        from oracle_lib import PEAK_DTYPE as _PK
        gpu_peaks = leg.d_peaks[(leg.step_no - 1) & 1][:min(n_tasks, 1024)].cpu().numpy().view(_PK).reshape(-1)

    soak_leg = None
    if args.soak_seconds > 0 and world == 1 and n_tasks > 0:
        soak_leg = soak(leg, cells_job, args.soak_seconds)

    # What one GPU's share of the capture costs at N = 8 (43 of the 340 runs = 1376 blocks per step, the one-rank all-reduce
    # included): a PREDICTION of the strong-scaling point the driver measures on an 8-GPU node, from this one GPU --
    # ms_per_step(whole capture) / ms_per_step(share) is the speed-up 8 GPUs reach if nothing but this step limits them.
    share = None
    if world == 1 and not grid and not iq8 and not args.capture and args.config == 1 and nblk >= 8 * 32 and args.soak_seconds > 0:
        share_runs = -(-(nblk // 32) // 8)
        sleg = Leg(torch, gpsacq, gdist, eng, dev, dist, backend, share_runs * 32, share_runs * 32, d_bits, None, stride, False)
        s_elapsed, s_kern_ms, _ = sleg.run(max(20, args.steps // 2), 3)
        s_ms = 1e3 * s_elapsed / max(20, args.steps // 2)
        s_tm = eng.last_timing()
        share = {"ranks_emulated": 8, "blocks_per_step": share_runs * 32, "ms_per_step": s_ms, "kernel_ms": s_kern_ms,
                 # the step's own three kernels (forward transforms, correlator, peak scan; HIP events) and what is left over: launch
                 # gaps, the key kernel, the event hand-over to torch's stream, the one-rank all-reduce
                 "stage_ms": {k: s_tm[k] for k in ("ms_sample", "ms_correlate", "ms_peaks")},
                 "non_kernel_ms": s_ms - (s_tm["ms_sample"] + s_tm["ms_correlate"] + s_tm["ms_peaks"]),
                 "cells_per_s_this_gpu": share_runs * 32 * eng.num_doppler / (s_ms * 1e-3),
                 "predicted_speedup_at_8_gpus": (1e3 * elapsed / args.steps) / s_ms,
                 "note": "one GPU running the largest per-rank share of the capture at N = 8 (same step, one-rank all-reduce included); "
                         "a prediction, not a measurement of 8 GPUs"}

    # What the one-rank process group costs the N = 1 line (rounds stay comparable): the same step without it, a few times
    no_coll = None
    if world == 1 and dist is not None and n_tasks > 0 and args.soak_seconds > 0:
        nleg = Leg(torch, gpsacq, gdist, eng, dev, None, backend, nblk, n_tasks, d_bits, d_tasks, stride, grid,
                   n_keys=(tasks.shape[0] if grid else 32), iq=iq_in)
        n_el, n_kern, _ = nleg.run(max(10, args.steps // 2), 2)
        no_coll = {"ms_per_step_without_process_group": 1e3 * n_el / max(10, args.steps // 2), "kernel_ms": n_kern,
                   "ms_per_step_with": 1e3 * elapsed / args.steps}
    # the fp32 vector pipe's own ceiling on this box, a few seconds under load (untimed)
    pk_stream, pk_clock = None, None
    if rank == 0 and world == 1 and args.pk_fma_seconds > 0 and args.soak_seconds > 0:
        leg.fence()
        smp = ClockSampler(torch, dev_index)
        smp.start()
        pk_stream = pk_fma_stream(args.pk_fma_seconds)
        smp.stop()
        pk_clock = smp.stats()

    weak = None
    if iq8:
        weak_blocks = 0
    if weak_blocks > 0:
        wleg = Leg(torch, gpsacq, gdist, eng, dev, dist, backend, weak_blocks, weak_blocks, make_capture(weak_blocks, 2000 + rank, 5120),
                   None, 5120, False)
        w_elapsed, w_kern_ms, _ = wleg.run(args.steps, args.warmup)
        w_cells = weak_blocks * eng.num_doppler
        weak = {"scaling": "weak", "blocks_per_gpu": weak_blocks, "value": w_cells * world * args.steps / w_elapsed, "unit": "cells/s",
                "ms_per_step": 1e3 * w_elapsed / args.steps, "kernel_ms": w_kern_ms,
                "kernel_cells_per_s": w_cells / (w_kern_ms * 1e-3) if w_kern_ms else None}

    parity_failed = False
    if rank == 0:
        value = cells_job * args.steps / elapsed
        fl = flops_per_cell(eng.num_lags)
        achieved_tf = cells_rank * fl / (kern_ms * 1e-3) / 1e12 if kern_ms else 0.0
        traffic, traffic_src, onchip, traffic_sha = None, None, None, None
        try:  # HBM bytes per launch and pipe utilisation from the committed PMC passes (profiles/traffic.json)
            tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
            traffic = (tj["hbm_read_bytes_per_cell"] + tj["hbm_write_bytes_per_cell"]) * cells_rank
            traffic_src = (f"profiles/{tj['tag']}_summary.md (rocprofv3 --pmc FETCH_SIZE, WRITE_SIZE in separate passes; "
                           "bytes per cell x cells per launch)")
            onchip = tj.get("onchip_counters")
            traffic_sha = tj.get("kernel_source_sha")
        except Exception:
            pass
        traffic_live = None
        if (world == 1 and not args.no_live_traffic and args.config == 1 and not iq8 and not args.capture
                and not any(k.startswith(("ROCP_", "ROCPROF")) for k in os.environ) and "rocprof" not in os.environ.get("LD_PRELOAD", "")):  # not under a profiler already
            # the default line: counters collected on THIS box, in this run (same capture size: the child generates the same data)
            traffic_live = live_traffic(["--blocks-total", str(args.blocks_total)])
            if "error" not in traffic_live:
                traffic = traffic_live["read_bytes"] + traffic_live["write_bytes"]
                traffic_src = ("live (estimated correction): rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate child runs of this command, "
                               "2 steps each), per k_corr launch; FETCH_SIZE KiB x 1024 x 2 (the guide's gfx950 correction for wide reads), WRITE_SIZE KiB x 1024 (uncalibrated)")
        sha_now = kernel_source_sha()
        # counters belong to the kernel binary they were collected on: flagged when the sources changed since (or the
        # profile predates the hash), or when this run's kernel instance is not the profiled one
        live_ok = traffic_live is not None and "error" not in traffic_live
        traffic_stale = False if live_ok else ((traffic is not None) and (traffic_sha != sha_now or args.config not in (1, 4) or iq8))
        alg_gbs = cells_rank * ALG_BYTES_PER_CELL / (kern_ms * 1e-3) / 1e9 if kern_ms else 0.0
        # Box-independent forms of the same measurement (the pool's boxes hold 2.09-2.20 GHz under this kernel): cycles one CU
        # spends per cell at the clock read DURING the timed steps, and the fraction of the peak at that clock (157.3 TFLOP/s = 2.4 GHz).
        sclk = clock["sclk_mhz"] if clock else None
        cyc_cell_cu = (kern_ms * 1e-3 * sclk * 1e6 * eng.compute_units / cells_rank) if (sclk and kern_ms and cells_rank) else None
        frac_at_clock = (achieved_tf / (FP32_VALU_PEAK_TF * sclk / FP32_PEAK_CLOCK_MHZ)) if (sclk and achieved_tf) else None
        pk_tf = pk_stream.get("pk_fma_stream_TF") if pk_stream else None
        out = {
            "metric": "(PRN,Doppler) correlation cells/s, 32 PRN @ fs=5.456 MHz" if args.config in (1, 4) else
                      f"(PRN,Doppler) correlation cells/s, 32 PRN @ fs={fs / 1e6:.3f} MHz",
            "value": value,
            "unit": "cells/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "capture file" if args.capture else "synthetic",
            "data_detail": (f"file {args.capture}" if args.capture else
                            ("device-generated 1-bit real-IF capture: white noise + PRNs %s at 45 dB-Hz, seeded Doppler/code phase" % injected)
                            if args.data == "signals" else "uniform random bits (sign of white noise)"),
            "config": {"workload": workload, "fs_hz": fs, "if_hz": cfg["fc"], "blocks_rank0": nblk,
                       "cells_per_step_rank0": cells_rank, "cells_per_step_job": cells_job, "parallelism": parallelism,
                       "input": "8-bit IQ (uint8 offset 128), converted inside the forward transform" if iq8 else "1-bit real IF"},
            "rccl_ranks_seen": ranks_seen, "dist_backend": (backend if dist is not None else None), "dist_note": dist_note,
            ("tasks_per_rank" if grid else "blocks_per_rank"): blocks_per_rank,
            # What binds k_corr is the fp32 vector pipe, not HBM: the fused kernel keeps the IFFT intermediate in LDS
            # and reads both spectra from L2, so the algorithmic bytes of SURVEY 8(d) never reach HBM.
            "roofline": {"bound": "valu_fp32", "kernel": f"k_corr<{eng.acc_columns}>", "achieved": achieved_tf,
                         "peak": FP32_VALU_PEAK_TF, "unit": "TFLOP/s", "frac": achieved_tf / FP32_VALU_PEAK_TF,
                         "flops_per_cell": fl, "flops_definition": "SURVEY.md 8(d): 6N + 5N log2 N + 5S",
                         "kernel_ms": kern_ms, "cells_per_launch": cells_rank,
                         "kernel_cells_per_s": cells_rank / (kern_ms * 1e-3) if kern_ms else None,
                         # clock / power read during the K timed steps; cycles = kernel_ms x sclk x CUs / cells
                         "sclk_mhz": sclk, "sclk_mhz_xcd_min": (clock.get("sclk_mhz_xcd_min_max") or [None])[0] if clock else None,
                         "sclk_mhz_xcd_max": (clock.get("sclk_mhz_xcd_min_max") or [None, None])[1] if clock else None,
                         "power_w": clock["power_w"] if clock else None, "compute_units": eng.compute_units,
                         "cycles_per_cell_per_cu": cyc_cell_cu, "frac_at_clock": frac_at_clock, "peak_clock_mhz": FP32_PEAK_CLOCK_MHZ,
                         # a pure v_pk_fma_f32 stream on this box (3 waves per SIMD like k_corr, a few seconds): the pipe's practical ceiling
                         "pk_fma_stream_TF": pk_tf, "frac_of_pk_fma_stream": (achieved_tf / pk_tf) if pk_tf else None,
                         "pk_fma_stream_sclk_mhz": pk_clock["sclk_mhz"] if pk_clock else None,
                         "clock_sampling": clock, "pk_fma_stream": pk_stream,
                         "traffic": traffic, "traffic_source": traffic_src, "traffic_stale": traffic_stale, "traffic_live": traffic_live,
                         "kernel_source_sha": sha_now, "traffic_kernel_source_sha": traffic_sha,
                         "hbm_secondary": {"algorithmic_bytes_per_cell": ALG_BYTES_PER_CELL, "algorithmic_GBs": alg_gbs,
                                           "hbm_peak_GBs": HBM_PEAK_GBS, "algorithmic_over_hbm_peak": alg_gbs / HBM_PEAK_GBS,
                                           "measured_hbm_GBs": (traffic / (kern_ms * 1e-3) / 1e9) if (traffic and kern_ms) else None,
                                           "note": "not a bound: the algorithmic bytes stay on chip; measured HBM traffic is what "
                                                   "`traffic` reports"},
                         "onchip_counters": onchip, "onchip_counters_stale": (onchip is not None) and (traffic_sha != sha_now),
                         "l2_peak_GBs": L2_PEAK_GBS},
            "stage_ms": {k: timing[k] for k in ("ms_total", "ms_sample", "ms_correlate", "ms_peaks")} if timing else None,
            "device": eng.device_name,
        }
        # which optional legs ran beside the K timed steps (none of them inside the timed region), so lines of different rounds
        # / command lines can be told apart; `one_rank_collective`: what the one-rank nccl group costs the N = 1 step
        out["extras"] = {"one_rank_process_group": dist is not None and world == 1, "soak": soak_leg is not None, "strong_share_at_8": share is not None,
                         "weak_scaling": weak is not None, "live_traffic": traffic_live is not None, "pk_fma_stream": pk_stream is not None,
                         "cpu_baseline": world == 1 and not args.no_cpu_baseline, "e2e_cli": world == 1 and not args.no_e2e,
                         "keys": "gpsacq_peak_keys_device (library, engine stream)"}
        if no_coll is not None:
            out["one_rank_collective"] = no_coll
        if soak_leg is not None:
            out["soak"] = soak_leg
        if share is not None:
            out["strong_share_at_8"] = share
        if weak is not None:
            out["weak_scaling"] = weak
        if iq8 and leg.sample_ms:
            ms = float(np.mean(leg.sample_ms))
            b_in, b_out = nblk * 80000, nblk * eng.doppler_sub * 8 * 5000 * 8
            out["ingest"] = {"kernel": "k_fwd2<iq8>", "ms": ms, "bytes_read": b_in, "bytes_written": b_out,
                             "GBs": (b_in + b_out) / (ms * 1e-3) / 1e9, "copy_ceiling_GBs": 6290.0,
                             "frac_of_copy_ceiling": (b_in + b_out) / (ms * 1e-3) / 1e9 / 6290.0,
                             "note": "8-bit IQ read (80 000 B per block) + polyphase spectrum written (320 KB per block) over the "
                                     "stage's HIP-event time; the stage also runs the mixer for 40 000 samples per block (float fast path, double-precision sincos where the sign could depend on it)"}
        if not grid:
            snr, lo, ca = gdist.unpack_keys(best.cpu(), eng.kmax)
            hits = torch.nonzero(snr >= 25).flatten().tolist()
            out["detected_prns"] = [int(p) + 1 for p in hits]
            if args.capture:
                out["detected"] = [{"prn": int(p) + 1, "snr": round(float(snr[p]), 1), "lo_shift": int(lo[p]), "ca_shift": int(ca[p])} for p in hits]
            elif args.data == "signals":
                # after the all-reduce the per-PRN best covers every rank's part of the capture (rank r: seed 1000 + r)
                out["injected_prns_all_ranks"] = sorted(set().union(*[synth_sats(1000 + r, fs)[0] for r in range(world)]))
        if world == 1 and not args.no_cpu_baseline:
            out["roofline"]["hbm_secondary"]["hbm_copy_measured_GBs"] = hbm_copy_gbs(torch, dev)
            host_bits = d_bits[:min(nblk, 1024) * 5120 if not grid else d_bits.numel()].cpu().numpy() if not iq8 else None
            ndop = eng.num_doppler_total if grid else eng.num_doppler
            if args.config in (1, 2) and not iq8:
                port = cpu_baseline(cfg, host_bits, ndop)
                ref = cpu_baseline_reference(cfg, host_bits, ndop)
                out["cpu_baseline"] = ref or port
                if ref:
                    out["cpu_baseline_port"] = port
                # the line's own parity verdict at the configuration it measures: the timed step's peaks vs the oracle's, same blocks
                if gpu_peaks is not None:
                    if args.parity_selftest:
                        gpu_peaks = gpu_peaks.copy()
                        i_bad = min(7, len(gpu_peaks) - 1)
                        gpu_peaks["ca_shift"][i_bad] = (gpu_peaks["ca_shift"][i_bad] + 1) % eng.num_lags
                    try:
                        par = parity_vs_gpu(cfg, eng, torch, d_bits, nblk, stride, gpu_peaks, cpu_baseline.last_peaks, host_bits)
                    except Exception as ex:
                        par = {"ok": False, "error": str(ex)[:300]}
                    cb = out["cpu_baseline"]
                    cb["parity_vs_gpu"] = par
                    # (flat copies: a consumer that keeps scalars only still sees the verdict)
                    cb.update({"parity_ok": par.get("ok"), "parity_blocks": par.get("blocks"), "parity_cells": par.get("cells"),
                               "parity_ca_equal": par.get("ca_equal"), "parity_lo_equal": par.get("lo_equal"), "parity_proven_ties": par.get("proven_ties"),
                               "parity_snr_max_rel": par.get("snr_max_rel"), "parity_pwr_max_rel": par.get("pwr_max_rel")})
                    parity_failed = not par.get("ok")
                try:
                    out["cpu_baseline_all_cores"] = cpu_baseline_all_cores(cfg, host_bits, ndop, port["value"])
                except Exception as ex:  # the 1-thread figure is the contract; this one is informative
                    out["cpu_baseline_all_cores"] = {"error": str(ex)}
        if world == 1 and not args.no_e2e and not grid and not iq8 and not args.capture and args.config in (1, 2):
            try:
                out["e2e_cli"] = e2e_cli(cfg, d_bits, nblk // 32, eng.num_doppler)
            except Exception as ex:
                out["e2e_cli"] = {"error": str(ex)}
        out.update(extra)
        emit(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()
    if parity_failed:  # the line is out (with the details); a GPU result the oracle contradicts is not a benchmark result
        print("bench.py: the timed step's results DISAGREE with the oracle (cpu_baseline.parity_vs_gpu)", file=sys.stderr)
        raise SystemExit(3)


if __name__ == "__main__":
    main()
