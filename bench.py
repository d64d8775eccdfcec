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
  --capture F  search the 1-bit capture file F (e.g. gps.samples.1bit.I.fs5456.if4092.bin) instead of
               synthetic data: whole runs of the file, same schedule, and the SearchTask report's hit list
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
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
L2_PEAK_GBS = 34500.0            # MI355X_MICROARCH.md: aggregate L2 bandwidth
ALG_BYTES_PER_CELL = 32 * N_FFT  # read signal + code spectra, write + read one IFFT intermediate

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
    cells, _ = orc.bench_blocks(bits[:nblk * 5120], nblk)
    dt = time.perf_counter() - t0
    return {"value": cells / dt, "unit": "cells/s", "cores": 1, "kind": "port",
            "sample": f"{nblk} blocks x {ndop} bins = {cells} cells of the same capture, oracle f32 build (own FFT, -O3), "
                      f"{dt:.1f} s on {os.cpu_count()} core host ({cpu_model()}), 1 thread"}


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


def cpu_baseline_all_cores(cfg, bits, ndop, target_s=8.0):
    """Same port on every core the process may use: one oracle instance per thread (ctypes releases the GIL), each
    searching chunks of 4 blocks of the same host sample until `target_s` seconds have passed (time-bounded: the cores a
    container really gets can be far fewer than it is shown)."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle_lib import Oracle
    ncpu = os.cpu_count() or 1
    try:
        ncpu = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        pass
    nblk = len(bits) // 5120
    chunk = 4

    def make(_):
        return Oracle(cfg["fc"], cfg["fs"], cfg["max_fo"], kind="f32")

    def work(i):
        cells, start = 0, (i * 7) % max(1, nblk - chunk + 1)
        while time.perf_counter() < deadline:
            cells += orcs[i].bench_blocks(bits[start * 5120:(start + chunk) * 5120], chunk)[0]
            start = (start + chunk) % max(1, nblk - chunk + 1)
        return cells

    with ThreadPoolExecutor(ncpu) as ex:
        orcs = list(ex.map(make, range(ncpu)))  # SearchInit() of every instance, untimed
        t0 = time.perf_counter()
        deadline = t0 + target_s
        cells = sum(ex.map(work, range(ncpu)))
        dt = time.perf_counter() - t0
    return {"value": cells / dt, "unit": "cells/s", "cores": ncpu, "kind": "port",
            "sample": f"{ncpu} threads (all cores the process may use) x chunks of {chunk} blocks x {ndop} bins for {target_s:.0f} s = {cells} cells, {dt:.1f} s"}


class Leg:
    """One timed workload: `n_tasks` tasks over `nblk` resident blocks on this rank."""

    def __init__(self, torch, gpsacq, gdist, eng, dev, dist, backend, nblk, n_tasks, d_bits, d_tasks, stride, grid, n_keys=32):
        self.torch, self.gdist, self.eng, self.dev, self.dist, self.backend = torch, gdist, eng, dev, dist, backend
        self.nblk, self.n_tasks, self.d_bits, self.d_tasks, self.stride, self.grid, self.n_keys = nblk, n_tasks, d_bits, d_tasks, stride, grid, n_keys
        # The search runs on the engine's own HIP stream; the key packing and the collective run on torch's
        # stream, ordered after it by an event, so step i's reduction / all-reduce overlaps step i+1's search
        # (two peak buffers; the engine stream waits for a buffer's previous reader).
        self.d_peaks = [torch.zeros((max(n_tasks, 1), 4), dtype=torch.int32, device=dev) for _ in range(2)]
        self.eng_stream = torch.cuda.ExternalStream(eng.stream_ptr, device=dev)
        self.reader_done = [None, None]
        self.step_no = 0

    def step(self):
        torch, eng = self.torch, self.eng
        slot = self.step_no & 1
        self.step_no += 1
        buf = self.d_peaks[slot]
        if self.n_tasks > 0:
            if self.reader_done[slot] is not None:
                self.eng_stream.wait_event(self.reader_done[slot])
            eng.search_device(self.d_bits.data_ptr(), self.nblk, buf.data_ptr(), stride=self.stride,
                              d_tasks_ptr=self.d_tasks.data_ptr() if self.d_tasks is not None else None,
                              n_tasks=self.n_tasks, sync=False)
            searched = torch.cuda.Event()
            searched.record(self.eng_stream)
            torch.cuda.current_stream().wait_event(searched)
        # best peak per PRN (block schedule) / per (block, PRN) (grid), packed so that integer MAX reproduces
        # the reference's ordering (higher SNR; ties -> lower Doppler bin, :198)
        # (a rank without work -- more ranks than runs / grid points -- contributes keys of 0, neutral for MAX)
        if self.n_tasks > 0:
            key = self.gdist.pack_keys(buf[:self.n_tasks], eng.kmax)
            best = key if self.grid else self.gdist.per_prn_best(key)
        else:
            best = torch.zeros(self.n_keys, dtype=torch.int64, device=self.dev)
        if self.dist is not None:
            if self.backend == "nccl":
                self.dist.all_reduce(best, op=self.dist.ReduceOp.MAX)  # RCCL over xGMI, 256 bytes
            else:
                b = best.cpu()
                self.dist.all_reduce(b, op=self.dist.ReduceOp.MAX)
                best = b.to(self.dev)
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
        corr_ms = []
        t0 = time.perf_counter()
        for i in range(steps):
            best = self.step()
            if i > 0 and self.n_tasks > 0:  # the previous search's times: waits for that search only
                corr_ms.append(self.eng.last_timing(1)["ms_correlate"])
        if self.n_tasks > 0:
            corr_ms.append(self.eng.last_timing(0)["ms_correlate"])
        self.fence()
        elapsed = time.perf_counter() - t0
        t = torch.tensor([elapsed], dtype=torch.float64, device=self.dev if self.backend == "nccl" else "cpu")
        if self.dist is not None:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item()), (float(np.mean(corr_ms)) if corr_ms else 0.0), best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", type=int, choices=[1, 2, 3, 4], default=1)
    ap.add_argument("--blocks-total", type=int, default=10880,
                    help="blocks of the whole capture (all ranks together; whole runs of 32): 10880 = the Nottingham capture")
    ap.add_argument("--weak-blocks", type=int, default=4096, help="blocks per rank of the weak-scaling leg (0: skip it)")
    ap.add_argument("--grid-blocks", type=int, default=4, help="--config 3/4: capture positions searched against all 32 PRNs")
    ap.add_argument("--doppler-step", type=float, default=0.0,
                    help="--config 3/4: requested Doppler step in Hz (0: the FFT bin fs/N); the engine takes the finest grid "
                         "it has that is not coarser (sub-bin phase ramps) or the coarsest not finer (bin stride)")
    ap.add_argument("--capture", default=None, help="1-bit capture file to search instead of synthetic data (config 1/2 schedules)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--data", choices=["signals", "noise"], default="signals",
                    help="signals (default): capture generated on the device, white noise + 8 PRNs at seeded Doppler / code "
                         "phase (SURVEY section 8d throughput set); noise: uniform random bits")
    args = ap.parse_args()

    import torch
    import gpsacq
    from gpsacq import dist as gdist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    # GPSACQ_DIST_BACKEND=gloo lets several ranks share one GPU to exercise the N > 1 code path on a 1-GPU box
    # (collectives then run on CPU copies); the driver's runs use nccl (= RCCL).
    backend = os.environ.get("GPSACQ_DIST_BACKEND", "nccl")
    dev_index = local_rank % torch.cuda.device_count() if backend == "gloo" else local_rank
    torch.cuda.set_device(dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend=backend)

    cfg = CONFIGS[args.config]
    grid = args.config in (3, 4)
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
        workload = (f"{cfg['name']}: 32 PRN x {eng.num_doppler} Doppler bins (+-5 kHz, fs/N = {fs / N_FFT:.1f} Hz), N=40000, "
                    f"{eng.num_lags} lags, reference schedule block->PRN (block % 32); capture of {total_runs * 32} blocks "
                    f"({total_runs} runs)" + (f" = file {os.path.basename(args.capture)}" if args.capture else ""))
        parallelism = f"whole runs split over {world} GPU(s) (strong scaling), per-PRN peak all-reduce(MAX) of 256 bytes"
        weak_blocks = 0 if args.capture else args.weak_blocks

    # input resident in HBM before anything is timed
    injected, sats = synth_sats(data_seed, fs)

    def make_capture(n_blocks, seed, blk_stride):
        if n_blocks == 0:
            return torch.zeros(5120, dtype=torch.uint8, device=dev)
        nbytes = (n_blocks - 1) * blk_stride + 5120
        if args.data == "signals":
            d = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            eng.generate_device(d.data_ptr(), nbytes, synth_sats(seed, fs)[1], noise_sigma=1.0, seed=seed)
            return d
        return torch.from_numpy(np.random.default_rng(seed).integers(0, 256, size=nbytes, dtype=np.uint8)).to(dev)

    if args.capture and not grid:
        with open(args.capture, "rb") as f:
            f.seek(first_run * 32 * 5120)
            host = np.frombuffer(f.read(nblk * 5120), dtype=np.uint8)
        d_bits = torch.from_numpy(host.copy()).to(dev) if nblk else torch.zeros(5120, dtype=torch.uint8, device=dev)
    else:
        d_bits = make_capture(nblk, data_seed, stride)

    leg = Leg(torch, gpsacq, gdist, eng, dev, dist, backend, nblk, n_tasks, d_bits, d_tasks, stride, grid,
              n_keys=(tasks.shape[0] if grid else 32))
    elapsed, kern_ms, best = leg.run(args.steps, args.warmup)
    timing = eng.last_timing() if n_tasks > 0 else None

    weak = None
    if weak_blocks > 0:
        wleg = Leg(torch, gpsacq, gdist, eng, dev, dist, backend, weak_blocks, weak_blocks, make_capture(weak_blocks, 2000 + rank, 5120),
                   None, 5120, False)
        w_elapsed, w_kern_ms, _ = wleg.run(args.steps, args.warmup)
        w_cells = weak_blocks * eng.num_doppler
        weak = {"scaling": "weak", "blocks_per_gpu": weak_blocks, "value": w_cells * world * args.steps / w_elapsed, "unit": "cells/s",
                "ms_per_step": 1e3 * w_elapsed / args.steps, "kernel_ms": w_kern_ms,
                "kernel_cells_per_s": w_cells / (w_kern_ms * 1e-3) if w_kern_ms else None}

    if rank == 0:
        value = cells_job * args.steps / elapsed
        fl = flops_per_cell(eng.num_lags)
        achieved_tf = cells_rank * fl / (kern_ms * 1e-3) / 1e12 if kern_ms else 0.0
        traffic, traffic_src, onchip = None, None, None
        try:  # HBM bytes per launch and pipe utilisation from the committed PMC passes (profiles/traffic.json)
            tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
            traffic = (tj["hbm_read_bytes_per_cell"] + tj["hbm_write_bytes_per_cell"]) * cells_rank
            traffic_src = (f"profiles/{tj['tag']}_summary.md (rocprofv3 --pmc FETCH_SIZE, WRITE_SIZE in separate passes; "
                           "bytes per cell x cells per launch)")
            onchip = tj.get("onchip_counters")
        except Exception:
            pass
        alg_gbs = cells_rank * ALG_BYTES_PER_CELL / (kern_ms * 1e-3) / 1e9 if kern_ms else 0.0
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
                       "cells_per_step_rank0": cells_rank, "cells_per_step_job": cells_job, "parallelism": parallelism},
            # What binds k_corr is the fp32 vector pipe, not HBM: the fused kernel keeps the IFFT intermediate in LDS
            # and reads both spectra from L2, so the algorithmic bytes of SURVEY 8(d) never reach HBM.
            "roofline": {"bound": "valu_fp32", "kernel": f"k_corr<{eng.acc_columns}>", "achieved": achieved_tf,
                         "peak": FP32_VALU_PEAK_TF, "unit": "TFLOP/s", "frac": achieved_tf / FP32_VALU_PEAK_TF,
                         "flops_per_cell": fl, "flops_definition": "SURVEY.md 8(d): 6N + 5N log2 N + 5S",
                         "kernel_ms": kern_ms, "cells_per_launch": cells_rank,
                         "kernel_cells_per_s": cells_rank / (kern_ms * 1e-3) if kern_ms else None,
                         "traffic": traffic, "traffic_source": traffic_src,
                         "hbm_secondary": {"algorithmic_bytes_per_cell": ALG_BYTES_PER_CELL, "algorithmic_GBs": alg_gbs,
                                           "hbm_peak_GBs": HBM_PEAK_GBS, "algorithmic_over_hbm_peak": alg_gbs / HBM_PEAK_GBS,
                                           "measured_hbm_GBs": (traffic / (kern_ms * 1e-3) / 1e9) if (traffic and kern_ms) else None,
                                           "note": "not a bound: the algorithmic bytes stay on chip; measured HBM traffic is what "
                                                   "`traffic` reports"},
                         "onchip_counters": onchip, "l2_peak_GBs": L2_PEAK_GBS},
            "stage_ms": {k: timing[k] for k in ("ms_total", "ms_sample", "ms_correlate", "ms_peaks")} if timing else None,
            "device": eng.device_name,
        }
        if weak is not None:
            out["weak_scaling"] = weak
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
            host_bits = d_bits[:min(nblk, 1024) * 5120 if not grid else d_bits.numel()].cpu().numpy()
            ndop = eng.num_doppler_total if grid else eng.num_doppler
            if args.config in (1, 2):
                port = cpu_baseline(cfg, host_bits, ndop)
                ref = cpu_baseline_reference(cfg, host_bits, ndop)
                out["cpu_baseline"] = ref or port
                if ref:
                    out["cpu_baseline_port"] = port
                try:
                    out["cpu_baseline_all_cores"] = cpu_baseline_all_cores(cfg, host_bits, ndop)
                except Exception as ex:  # the 1-thread figure is the contract; this one is informative
                    out["cpu_baseline_all_cores"] = {"error": str(ex)}
        out.update(extra)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
