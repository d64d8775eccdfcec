#!/usr/bin/env python3
"""bench.py -- (PRN, Doppler) correlation cells/s of the MI355X acquisition engine: the timed core.  Everything reported BESIDE the K
timed steps lives in bench_extras.py.

Contract: `python bench.py --gpus N --steps K --warmup W`; for N > 1 the driver launches one rank per GPU through
torch.distributed.run (RCCL).  Rank 0 prints ONE JSON line.  `--gpus N` without WORLD_SIZE re-launches itself with N ranks.

Workload (BASELINE.json configs[1]): 32 PRN, fs = 5.456 MHz, IF = 4.092 MHz, N = 40000, +-5 kHz -> 73 Doppler bins of fs/N, 5456 lags;
reference schedule (SearchTask, c/search_offline.cpp:239-246): block b of the capture against PRN b % 32 over all bins.  One step = one
pass of the whole hot path (unpack + mix + forward FFT-40000 per block, 73 fused multiply / IFFT-40000 / peak cells per block, the
per-(block, PRN) peak, the per-PRN merge keys, ONE all-reduce(MAX) of 256 bytes) over a capture of `--blocks-total` blocks resident in
HBM: 10 880 blocks = 340 runs, the size of the Nottingham capture the metric is quoted on.

Scaling: STRONG, and every world size searches THE SAME capture.  The capture is a function of (seed, absolute sample index) alone
(gpsacq_generate_range_device): rank r of N generates exactly blocks [first_run * 32, ...) of the stream the N = 1 job searches
(gpsacq.dist.shard_runs).  `keys_digest` -- sha256 of the 32 merged per-PRN keys of the last timed step (snr, Doppler bin, code phase
of the best peak of every PRN over the whole capture; c/search_offline.cpp:196-198) -- and `detected` are therefore IDENTICAL at N = 1,
2, 4, 8 by construction; rank 0's WHOLE share of the last timed step -- every block's peak, every cell -- is checked against the oracle at
any N (cpu_baseline.parity_vs_gpu / parity_vs_gpu: the oracle's all-cores pass, which is also the cpu_baseline_all_cores figure); at N > 1
rank 0 also drives all N devices once through the C ABI's own gpsacq_multi_search_blocks (extras.inproc_multi) and compares its keys.

Other lines (the default stays the one the driver records): --config 2|3|4 (BASELINE configs[2..4]), --input iq8 (8-bit IQ capture
converted inside the forward transform), --capture FILE (a 1-bit capture file, e.g. gps.samples.1bit.I.fs5456.if4092.bin).
--bare: the timed steps only (what the profiling and PMC child runs use).  Exit code 3: the GPU's results disagree with the oracle.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "gnss-gps-sdr_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import bench_extras as X  # noqa: E402
from bench_extras import ClockSampler, compare_peaks, cu_clocks, kernel_source_sha, read_clock_power  # noqa: E402,F401  (tests, tools/profile.sh)

from bench_extras import (ALG_BYTES_PER_CELL, CAPTURE_SEED, CONFIGS, FP32_PEAK_CLOCK_MHZ, FP32_VALU_PEAK_TF, HBM_PEAK_GBS, L2_PEAK_GBS, N_FFT,  # noqa: E402,F401
                          STAMP_SLOTS, flops_per_cell, make_iq_capture, synth_sats)

class Leg:
    """One timed workload: `n_tasks` tasks over `nblk` resident blocks on this rank."""

    def __init__(self, torch, eng, dev, dist, backend, nblk, n_tasks, d_bits, d_tasks, stride, grid, n_keys=32, iq=None, keep_cells=False):
        self.torch, self.eng, self.dev, self.dist, self.backend = torch, eng, dev, dist, backend
        self.iq = iq  # gpsacq.Iq8Input: d_bits then holds interleaved 8-bit I,Q bytes, `stride` bytes per block
        self.nblk, self.n_tasks, self.d_bits, self.d_tasks, self.stride, self.grid, self.n_keys = nblk, n_tasks, d_bits, d_tasks, stride, grid, n_keys
        # The search and the key packing run on the engine's own HIP stream, the collective on torch's stream behind an event, so step
        # i's all-reduce overlaps the searches behind it.  THREE peak / key buffers (the engine stream waits for a buffer's previous
        # reader): the correlate kernel's workgroups are persistent and fill every CU until the kernel ends, so the all-reduce kernel of
        # step i gets its CU at the boundary between the correlators of steps i+1 and i+2 -- with two buffers step i+2's search would
        # wait for it there (a rendezvous of all ranks in every step), with three it has until step i+3.
        self.NBUF = 3
        self.d_peaks = [torch.zeros((max(n_tasks, 1), 4), dtype=torch.int32, device=dev) for _ in range(self.NBUF)]
        self.d_keys = [torch.zeros(n_keys, dtype=torch.int64, device=dev) for _ in range(self.NBUF)]
        # keep_cells: every step writes its cells (16 bytes per cell, the kernel's only per-cell output) into this buffer instead of the
        # engine's own scratch -- the same kernels, the same launches -- so that the LAST TIMED STEP's cells can be checked afterwards
        self.d_cells = torch.empty((n_tasks, eng.num_doppler, 4), dtype=torch.int32, device=dev) if (keep_cells and n_tasks > 0 and iq is None) else None
        self.sampler = None  # bench_extras.ClockSampler: sclk / power readings + cycle stamps during the timed steps of run()
        self.eng_stream = torch.cuda.ExternalStream(eng.stream_ptr, device=dev)
        self.reader_done = [None] * self.NBUF
        self.step_no = 0

    def step(self):
        torch, eng = self.torch, self.eng
        slot = self.step_no % self.NBUF
        self.step_no += 1
        buf, best = self.d_peaks[slot], self.d_keys[slot]
        if self.n_tasks > 0:
            if self.reader_done[slot] is not None:
                self.eng_stream.wait_event(self.reader_done[slot])
            if self.iq is not None:
                eng.search_iq8_device(self.d_bits.data_ptr(), self.iq, self.nblk, buf.data_ptr(), stride=self.stride, sync=False)
            else:
                eng.search_device(self.d_bits.data_ptr(), self.nblk, buf.data_ptr(), stride=self.stride,
                                  d_tasks_ptr=self.d_tasks.data_ptr() if self.d_tasks is not None else None, n_tasks=self.n_tasks,
                                  d_cells_ptr=self.d_cells.data_ptr() if self.d_cells is not None else None, sync=False)
            # best peak per PRN (block schedule) / per (block, PRN) (grid) as keys whose integer MAX is the reference's ordering (higher SNR;
            # ties -> lower Doppler bin, :198): made by the library, one launch on the engine's stream.  Every key of the buffer is rewritten
            eng.peak_keys_device(buf.data_ptr(), self.n_tasks, best.data_ptr(), per_prn=not self.grid, sync=False)
            searched = torch.cuda.Event()
            searched.record(self.eng_stream)
            torch.cuda.current_stream().wait_event(searched)
        else:
            best.zero_()  # a rank without work contributes zeros (neutral for MAX) EVERY step: the buffer holds an earlier step's merged keys
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
        """W untimed + K timed steps bracketed by barrier + synchronize; returns (max-over-ranks seconds, mean correlate-kernel ms on
        this rank, last best keys)."""
        torch = self.torch
        best = None
        for _ in range(warmup):
            best = self.step()
        stamps = ev = None
        if self.sampler is not None and self.n_tasks > 0:  # allocated and zeroed BEFORE the fence: ordered before the engine stream's stamp kernel
            stamps = torch.zeros((2, STAMP_SLOTS), dtype=torch.int64, device=self.dev)  # [before / after][xcc << 6 | se << 4 | cu]
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        self.fence()
        corr_ms, self.sample_ms = [], []
        if self.sampler is not None:
            self.sampler.start()
            if stamps is not None:  # shader-cycle stamps + HIP events on the engine's stream around the timed steps: the clock the GPU itself counted
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
        if stamps is not None:  # the one launch inside the timed region that is not a step: 4096 tiny workgroups, its own time reported
            ev[1].record(self.eng_stream)
            self.eng.cycle_stamp_device(stamps[1].data_ptr())
            ev[2].record(self.eng_stream)
        self.fence()
        elapsed = time.perf_counter() - t0
        self.memtime_mhz = self.memtime_per_xcd = self.memtime_cus = self.stamp_kernel_ms = None
        if stamps is not None:
            self.stamp_kernel_ms = ev[1].elapsed_time(ev[2])
            self.memtime_mhz, self.memtime_per_xcd, self.memtime_cus = cu_clocks(stamps.cpu().numpy(), ev[0].elapsed_time(ev[2]))
        if self.sampler is not None:
            self.sampler.stop()
        t = torch.tensor([elapsed], dtype=torch.float64, device=self.dev if self.backend == "nccl" else "cpu")
        if self.dist is not None:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item()), (float(np.mean(corr_ms)) if corr_ms else 0.0), (best.clone() if best is not None else None)


_REAL_STDOUT = None


def claim_stdout():
    """From here on file descriptor 1 is stderr, and the JSON line is written to the saved descriptor of the real stdout: libraries that
    print to the C-level stdout (RCCL's version banner, flushed at exit) can no longer add lines to what the driver parses."""
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.dup(1)
        os.dup2(2, 1)


def emit(line):
    if _REAL_STDOUT is None:
        sys.stdout.write(line + "\n")
        sys.stdout.flush()
    else:
        os.write(_REAL_STDOUT, (line + "\n").encode())


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)  # 100 x 42 ms: a GPU leg long enough for a 5-second SMI sampler to see it
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, choices=[1, 2, 3, 4], default=1)
    ap.add_argument("--blocks-total", type=int, default=10880, help="blocks of the whole capture (all ranks together; whole runs of 32): 10880 = the Nottingham capture")
    ap.add_argument("--weak-blocks", type=int, default=4096, help="blocks per rank of the weak-scaling leg (0: skip it)")
    ap.add_argument("--grid-blocks", type=int, default=4, help="--config 3/4: capture positions searched against all 32 PRNs")
    ap.add_argument("--doppler-step", type=float, default=0.0, help="--config 3/4: requested Doppler step in Hz (0: the FFT bin fs/N)")
    ap.add_argument("--capture", default=None, help="1-bit capture file to search instead of synthetic data (config 1/2 schedules)")
    ap.add_argument("--input", choices=["bits", "iq8"], default="bits", help="iq8: an 8-bit IQ capture (uint8, offset 128) converted inside the forward transform")
    ap.add_argument("--data", choices=["signals", "noise"], default="signals", help="signals: white noise + 8 PRNs at seeded Doppler / code phase, generated on the device; noise: uniform random bits")
    ap.add_argument("--bare", action="store_true", help="the timed steps only: no weak leg, soak, CPU baseline, live traffic, library baseline, e2e, in-process multi leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-dist", action="store_true", help="N = 1: do not create the one-rank nccl process group")
    ap.add_argument("--force-dist", action="store_true", help="N = 1: the one-rank nccl group must come up (no fallback)")
    ap.add_argument("--soak-seconds", type=float, default=6.0, help="GPU time of the soak leg after the timed steps (0: skip it and the other N = 1 side legs)")
    ap.add_argument("--live-traffic", action="store_true", help="(kept for old command lines: live traffic is on unless --no-live-traffic)")
    ap.add_argument("--no-live-traffic", action="store_true", help="do not measure roofline.traffic in this run (two rocprofv3 --pmc child runs); use profiles/traffic.json")
    ap.add_argument("--no-library-baseline", action="store_true", help="skip extras.gpu_library_baseline (the same cells through rocFFT)")
    ap.add_argument("--no-inproc-multi", action="store_true", help="N > 1: skip the in-process gpsacq_multi_search_blocks leg")
    ap.add_argument("--parity-selftest", action="store_true", help="corrupt one GPU peak (ca_shift + 1) before the parity verdict: the run must then exit 3 (tests)")
    ap.add_argument("--parity-blocks", type=int, default=0, help="N > 1: blocks of rank 0's share pushed through the oracle for the parity verdict (0: the whole share)")
    ap.add_argument("--parity-seconds", type=float, default=60.0, help="time limit of the oracle's all-cores pass over rank 0's share (blocks it does not reach are not part of the verdict)")
    ap.add_argument("--pk-fma-seconds", type=float, default=3.0, help="N = 1: seconds of the pure v_pk_fma_f32 stream behind roofline.pk_fma_stream_TF (0: skip)")
    ap.add_argument("--no-e2e", action="store_true", help="skip the gps_test end-to-end leg")
    ap.add_argument("--spawn-check", action="store_true", help="launcher check without a GPU: every rank joins a gloo group, rank 0 prints the ranks it saw")
    a = ap.parse_args()
    if a.bare:
        a.weak_blocks, a.soak_seconds, a.pk_fma_seconds = 0, 0.0, 0.0
        a.no_cpu_baseline = a.no_live_traffic = a.no_library_baseline = a.no_inproc_multi = a.no_e2e = True
    return a


def init_dist(args, torch, backend, world, dev_index):
    """The process group: N > 1 as launched; N = 1 a ONE-RANK nccl group, so that every collective line of the N > 1 runs executes in
    the driver's N = 1 line too.  Returns (dist or None, note, host-side group or None): the ranks that have nothing left to do wait for
    rank 0's untimed legs in a gloo barrier -- on the CPU, their GPUs idle -- not inside an RCCL kernel."""
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev_index))
            host_group = None
            try:
                host_group = dist.new_group(backend="gloo")
            except Exception as ex:  # (no usable interface for gloo: the final wait then uses the RCCL barrier)
                print(f"bench.py: no gloo group for the final wait ({str(ex)[:120]})", file=sys.stderr)
            ok = torch.tensor([1 if host_group is not None else 0], dtype=torch.int32, device=torch.device("cuda", dev_index))
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)  # every rank takes the same branch
            return dist, None, (host_group if int(ok.item()) == 1 else None)
        dist.init_process_group(backend=backend)
        return dist, None, None
    if args.no_dist or backend != "nccl":
        return None, None, None
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
        return dist, None, None
    except Exception as ex:
        if args.force_dist:
            raise
        try:
            if dist.is_initialized():
                dist.destroy_process_group()
        except Exception:
            pass
        return None, f"one-rank nccl group unavailable ({str(ex)[:160]}): ran without a process group", None


def main():
    args = parse_args()
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
    rank, local_rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
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
    # GPSACQ_DIST_BACKEND=gloo lets several ranks share one GPU to exercise the N > 1 code path on a 1-GPU box (collectives then run on
    # CPU copies); the driver's runs use nccl (= RCCL).
    dev_index = local_rank % torch.cuda.device_count() if backend == "gloo" else local_rank
    if dev_index >= torch.cuda.device_count():  # more ranks than GPUs on the RCCL backend: never share a device silently
        raise SystemExit(f"rank {rank}: device {dev_index} not visible ({torch.cuda.device_count()} device(s), --gpus {args.gpus})")
    torch.cuda.set_device(dev_index)
    dist, dist_note, host_group = init_dist(args, torch, backend, world, dev_index)
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
    fs, stride, first_run, total_runs, tasks = cfg["fs"], 5120, 0, 0, None

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
        cells_rank, cells_job = tasks.shape[0] * nbins * n_acc, tasks.shape[0] * total_bins * n_acc
        workload = (f"{cfg['name']}: {npos} capture positions x 32 PRN x {total_bins} Doppler points (+-{cfg['max_fo'] / 1e3:.0f} kHz, step "
                    f"{eng.doppler_step_hz:.2f} Hz), N=40000, {eng.num_lags} lags" + (f", {n_acc} non-coherent sums (blocks {stride} bytes apart)" if n_acc > 1 else ""))
        parallelism = f"Doppler slabs over {world} GPU(s), per-(position, PRN) peak all-reduce(MAX)"
        weak_blocks = 0
    else:
        total_runs = os.path.getsize(args.capture) // (32 * 5120) if args.capture else max(1, args.blocks_total // 32)  # (SearchTask stops at the first short read, :241-244)
        first_run, n_runs = gdist.shard_runs(total_runs, rank, world)
        nblk = n_runs * 32
        data_seed = CAPTURE_SEED + rank if iq8 else CAPTURE_SEED  # (the IQ stand-in comes from a sequential torch generator: one stream per rank)
        d_tasks, n_tasks = None, nblk
        cells_rank, cells_job = nblk * eng.num_doppler, total_runs * 32 * eng.num_doppler
        workload = (f"{cfg['name']}: 32 PRN x {eng.num_doppler} Doppler bins (+-{cfg['max_fo'] / 1e3:.0f} kHz, fs/N = {fs / N_FFT:.1f} Hz), N=40000, "
                    f"{eng.num_lags} lags, reference schedule block->PRN (block % 32); capture of {total_runs * 32} blocks ({total_runs} runs)"
                    + (f" = file {os.path.basename(args.capture)}" if args.capture else ""))
        parallelism = f"whole runs of ONE capture split over {world} GPU(s) (strong scaling), per-PRN peak all-reduce(MAX) of 256 bytes"
        weak_blocks = 0 if (args.capture or iq8) else args.weak_blocks

    # input resident in HBM before anything is timed
    injected, sats = synth_sats(data_seed, fs)

    def make_capture(n_blocks, seed, blk_stride, first_block=0):
        """Blocks [first_block, first_block + n_blocks) of the capture `seed` (a function of the absolute sample index)."""
        if n_blocks == 0:
            return torch.zeros(5120, dtype=torch.uint8, device=dev)
        nbytes = (n_blocks - 1) * blk_stride + 5120
        if args.data == "signals":
            d = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            torch.cuda.synchronize()
            eng.generate_device(d.data_ptr(), nbytes, synth_sats(seed, fs)[1], noise_sigma=1.0, seed=seed, first_sample=first_block * blk_stride * 8)
            return d
        whole = np.random.default_rng(seed).integers(0, 256, size=(first_block + n_blocks) * blk_stride, dtype=np.uint8)
        return torch.from_numpy(whole[first_block * blk_stride:first_block * blk_stride + nbytes].copy()).to(dev)

    iq_in = None
    if iq8:
        stride = 81920
        d_bits, iq_mean = make_iq_capture(torch, dev, fs, max(nblk, 1), data_seed)
        iq_in = eng.iq8_input(signed=False, remove_dc=True, mean=iq_mean, mix_hz=cfg["fc"], fs=fs, first_sample=0, total_samples=max(nblk, 1) * 40960)
    elif args.capture and not grid:
        with open(args.capture, "rb") as f:
            f.seek(first_run * 32 * 5120)
            host = np.frombuffer(f.read(nblk * 5120), dtype=np.uint8)
        d_bits = torch.from_numpy(host.copy()).to(dev) if nblk else torch.zeros(5120, dtype=torch.uint8, device=dev)
    else:
        d_bits = make_capture(nblk, data_seed, stride, first_block=0 if grid else first_run * 32)

    # every rank's share of the work and its device, gathered before anything is timed (what SCALE lines are checked against)
    blocks_per_rank, devices = [nblk], [dev_index]
    if dist is not None:
        t = torch.zeros((world, 2), dtype=torch.int64, device=dev if backend == "nccl" else "cpu")
        t[rank, 0], t[rank, 1] = (n_tasks if grid else nblk), dev_index
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        blocks_per_rank, devices = [int(v) for v in t[:, 0].tolist()], [int(v) for v in t[:, 1].tolist()]

    n_keys = tasks.shape[0] if grid else 32
    # rank 0's share of the LAST TIMED STEP is checked against the oracle afterwards (peaks and cells): its cells are kept
    check_parity = rank == 0 and not grid and not iq8 and n_tasks > 0 and not args.no_cpu_baseline and args.config in (1, 2)
    leg = Leg(torch, eng, dev, dist, backend, nblk, n_tasks, d_bits, d_tasks, stride, grid, n_keys=n_keys, iq=iq_in, keep_cells=check_parity)
    if rank == 0:
        leg.sampler = ClockSampler(torch, dev_index)  # sclk / power during the K timed steps (roofline.sclk_mhz, .power_w)
    elapsed, kern_ms, best = leg.run(args.steps, args.warmup)
    timing = eng.last_timing() if n_tasks > 0 else None
    clock = X.clock_of_leg(leg) if leg.sampler is not None else None
    leg.sampler = None
    ms_step = 1e3 * elapsed / args.steps
    # the peaks and cells of the LAST TIMED STEP (this rank's whole share), copied out for the parity verdict before any other leg runs
    gpu_peaks = gpu_cells = None
    if check_parity:
        from oracle_lib import CELL_DTYPE as _CL, PEAK_DTYPE as _PK
        gpu_peaks = leg.d_peaks[(leg.step_no - 1) % leg.NBUF][:n_tasks].cpu().numpy().view(_PK).reshape(-1)
        gpu_cells = leg.d_cells.cpu().numpy().view(_CL).reshape(n_tasks, eng.num_doppler)

    side = world == 1 and n_tasks > 0 and args.soak_seconds > 0  # the N = 1 side legs
    soak_leg = X.soak(leg, cells_job, args.soak_seconds) if side else None
    share = X.strong_share_at_8(Leg, leg, ms_step, args.steps) if (side and not grid and not iq8 and not args.capture and args.config == 1 and nblk >= 256) else None
    no_coll = X.one_rank_collective(Leg, leg, ms_step, args.steps) if (side and dist is not None) else None
    pk_stream, pk_clock = None, None
    if side and rank == 0 and args.pk_fma_seconds > 0:  # the fp32 vector pipe's own ceiling on this box, a few seconds under load
        leg.fence()
        smp = ClockSampler(torch, dev_index)
        smp.start()
        pk_stream = X.pk_fma_stream(args.pk_fma_seconds)
        smp.stop()
        pk_clock = smp.stats()
    weak = None
    if weak_blocks > 0:
        wleg = Leg(torch, eng, dev, dist, backend, weak_blocks, weak_blocks, make_capture(weak_blocks, 2000 + rank, 5120), None, 5120, False)
        w_elapsed, w_kern_ms, _ = wleg.run(args.steps, args.warmup)
        w_cells = weak_blocks * eng.num_doppler
        weak = {"scaling": "weak", "blocks_per_gpu": weak_blocks, "value": w_cells * world * args.steps / w_elapsed, "unit": "cells/s",
                "ms_per_step": 1e3 * w_elapsed / args.steps, "kernel_ms": w_kern_ms, "kernel_cells_per_s": w_cells / (w_kern_ms * 1e-3) if w_kern_ms else None}

    parity_failed = False
    if rank == 0:
        import types
        out, parity_failed = X.build_line(types.SimpleNamespace(**locals()))
        emit(json.dumps(out))
    if dist is not None:
        if world > 1:  # the other ranks wait here, on the host, while rank 0 runs its untimed legs (oracle, in-process multi-GPU leg)
            torch.cuda.synchronize()
            dist.barrier(group=host_group) if host_group is not None else dist.barrier()
        dist.destroy_process_group()
    if parity_failed:  # the line is out (with the details); a GPU result the oracle contradicts is not a benchmark result
        print("bench.py: the timed step's results DISAGREE with the oracle (parity_vs_gpu)", file=sys.stderr)
        raise SystemExit(3)


if __name__ == "__main__":
    main()
