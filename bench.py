#!/usr/bin/env python3
"""bench.py -- (PRN, Doppler) correlation cells/s of the MI355X acquisition engine.

Contract (see the task statement): `python bench.py --gpus N --steps K --warmup W`; for N > 1
the driver launches one rank per GPU through torch.distributed.run (RCCL).  Rank 0 prints ONE
JSON line.

Workload (BASELINE.json configs[1]): 32 PRN, fs = 5.456 MHz, IF = 4.092 MHz, N = 40000,
+-5 kHz -> 73 Doppler bins of fs/N, 5456 lags scanned; reference schedule (SearchTask,
c/search_offline.cpp:239-246): every 5120-byte block of the capture is searched against PRN
(block % 32) over all Doppler bins.  One step = one pass of the whole hot path (1-bit unpack +
mix + forward FFT-40000 per block, then 73 fused multiply/IFFT-40000/peak cells per block,
then the per-(block, PRN) peak) over `--blocks` synthetic blocks already resident in HBM, plus
-- for N > 1 -- one RCCL all-reduce(MAX) of the 32 per-PRN best peaks (256 bytes).
Weak scaling: every rank searches its own `--blocks` blocks.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "gnss-gps-sdr_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

FC, FS, MAX_FO = 4.092e6, 5.456e6, 5000.0
N_FFT = 40000
ALG_BYTES_PER_CELL = 32 * N_FFT  # read signal + code spectra, write + read one IFFT intermediate
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def synth_bits(n_blocks, seed):
    """Synthetic 1-bit real-IF capture: sign bits of white noise (uniform random bits)."""
    rng = np.random.default_rng(seed)
    return rng.integers(0, 256, size=n_blocks * 5120, dtype=np.uint8)


def synth_sats(seed):
    """The 8 satellites injected into the capture generated from `seed`: (sorted PRNs, generator tuples)."""
    rs = np.random.default_rng(seed)
    prns = sorted(rs.choice(np.arange(1, 33), size=8, replace=False).tolist())
    return prns, [(prn, 0.151, float(rs.uniform(-4500, 4500)), float(rs.uniform(0, 5456)), float(rs.random())) for prn in prns]


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


def hbm_copy_gbs(torch, dev, nbytes=1 << 30, reps=5):
    """Device-to-device copy rate (read + write bytes per second) as the measured counterpart of the
    8 TB/s vendor peak (SURVEY.md section 8d asks for both)."""
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


def cpu_baseline(bits, target_s=12.0):
    """The oracle's float build (own mixed-radix FFT; `port`) timed single-threaded on a
    bounded sample of the same workload."""
    from oracle_lib import Oracle
    orc = Oracle(FC, FS, MAX_FO, kind="f32")
    t0 = time.perf_counter()
    cells, _ = orc.bench_blocks(bits[:2 * 5120], 2)
    dt = time.perf_counter() - t0
    nblk = int(max(2, min(len(bits) // 5120, target_s / (dt / 2))))
    t0 = time.perf_counter()
    cells, _ = orc.bench_blocks(bits[:nblk * 5120], nblk)
    dt = time.perf_counter() - t0
    return {"value": cells / dt, "unit": "cells/s", "cores": 1, "kind": "port",
            "sample": f"{nblk} blocks x 73 bins = {cells} cells of the same capture, oracle f32 build (own FFT, -O3), "
                      f"{dt:.1f} s on {os.cpu_count()} core host ({cpu_model()}), 1 thread"}


def cpu_baseline_all_cores(bits, single_rate, target_s=8.0):
    """Same port, one oracle instance per host core (threads; ctypes releases the GIL) -- the
    'all cores' figure SURVEY.md section 8(d) asks for next to the 1-thread one."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle_lib import Oracle
    ncpu = min(os.cpu_count() or 1, 64)
    per = int(max(1, min(len(bits) // 5120 // ncpu, target_s * single_rate / 73)))
    orcs = [Oracle(FC, FS, MAX_FO, kind="f32") for _ in range(ncpu)]

    def work(i):
        return orcs[i].bench_blocks(bits[i * per * 5120:(i + 1) * per * 5120], per)[0]

    t0 = time.perf_counter()
    with ThreadPoolExecutor(ncpu) as ex:
        cells = sum(ex.map(work, range(ncpu)))
    dt = time.perf_counter() - t0
    return {"value": cells / dt, "unit": "cells/s", "cores": ncpu, "kind": "port",
            "sample": f"{ncpu} threads x {per} blocks x 73 bins = {cells} cells, {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--blocks", type=int, default=4096, help="5120-byte blocks per GPU per step (128 runs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--mode", choices=["blocks", "grid"], default="blocks",
                    help="blocks (default, the metric's config): reference schedule, blocks sharded over ranks, weak scaling. "
                         "grid (BASELINE configs[4]): --grid-blocks blocks x 32 PRN x +-100 kHz fine grid, Doppler slabs "
                         "sharded over ranks, per-(block, PRN) peak all-reduce, strong scaling")
    ap.add_argument("--grid-blocks", type=int, default=4)
    ap.add_argument("--data", choices=["signals", "noise"], default="signals",
                    help="signals (default): capture generated on the device, white noise + 8 PRNs at seeded Doppler / code "
                         "phase (SURVEY section 8d throughput set); noise: host-generated random bits")
    args = ap.parse_args()

    import torch
    import gpsacq

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    # GPSACQ_DIST_BACKEND=gloo lets two ranks share one GPU to smoke-test the N > 1 code path on a
    # 1-GPU box (collectives then run on CPU copies); the driver's runs use nccl (= RCCL).
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

    from gpsacq import dist as gdist
    grid = args.mode == "grid"
    eng = gpsacq.Engine(FC, FS, 100000.0 if grid else MAX_FO, device=dev_index)
    dev = torch.device("cuda", dev_index)
    if grid:
        # every rank holds the same few blocks, searches all 32 PRNs over ITS slab of Doppler bins
        nblk = args.grid_blocks
        data_seed = 77
        first, nbins = gdist.shard_doppler(eng.dmax, rank, world)
        total_bins = 2 * eng.dmax + 1
        eng.set_doppler_window(first, nbins)
        tasks = np.array([(b, sv) for b in range(nblk) for sv in range(32)], dtype=np.int32)
        d_tasks = torch.from_numpy(tasks).to(dev)
        n_tasks = tasks.shape[0]
        cells_per_step = n_tasks * nbins          # this rank's share
        job_cells_per_step = n_tasks * total_bins  # whole job, fixed as N grows
    else:
        nblk = args.blocks
        data_seed = 1000 + rank
        d_tasks, n_tasks = None, nblk
        cells_per_step = nblk * eng.num_doppler
        job_cells_per_step = cells_per_step * world
    # synthetic input, resident in HBM before anything is timed
    injected, sats = synth_sats(data_seed)
    if args.data == "signals":
        d_bits = torch.empty(nblk * 5120, dtype=torch.uint8, device=dev)
        eng.generate_device(d_bits.data_ptr(), nblk * 5120, sats, noise_sigma=1.0, seed=data_seed)
        host_bits = d_bits.cpu().numpy()
    else:
        host_bits = synth_bits(nblk, data_seed)
        d_bits = torch.from_numpy(host_bits).to(dev)
    # The search runs on the engine's own HIP stream; the peak reduction and the collective run on
    # torch's stream, ordered after it by an event, so step i's reduction / all-reduce overlaps step
    # i+1's search (two peak buffers; the engine stream waits for a buffer's previous reader).
    d_peaks = [torch.zeros((n_tasks, 4), dtype=torch.int32, device=dev) for _ in range(2)]
    eng_stream = torch.cuda.ExternalStream(eng.stream_ptr, device=dev)
    reader_done = [None, None]
    step_no = [0]

    def step():
        slot = step_no[0] & 1
        step_no[0] += 1
        buf = d_peaks[slot]
        if reader_done[slot] is not None:
            eng_stream.wait_event(reader_done[slot])
        eng.search_device(d_bits.data_ptr(), nblk, buf.data_ptr(), d_tasks_ptr=d_tasks.data_ptr() if grid else None,
                          n_tasks=n_tasks, sync=False)
        searched = torch.cuda.Event()
        searched.record(eng_stream)
        torch.cuda.current_stream().wait_event(searched)
        # best peak per PRN (blocks mode) / per (block, PRN) (grid mode), packed so that integer MAX
        # reproduces the reference's ordering (higher SNR; ties -> lower Doppler bin, :198)
        key = gdist.pack_keys(buf, eng.dmax)
        best = key if grid else gdist.per_prn_best(key)
        if dist is not None:
            if backend == "nccl":
                dist.all_reduce(best, op=dist.ReduceOp.MAX)  # RCCL over xGMI, 256 bytes
            else:
                b = best.cpu()
                dist.all_reduce(b, op=dist.ReduceOp.MAX)
                best = b.to(dev)
        reader_done[slot] = torch.cuda.Event()
        reader_done[slot].record(torch.cuda.current_stream())
        return best

    def fence():
        eng.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    corr_ms = []
    t0 = time.perf_counter()
    for i in range(args.steps):
        best = step()
        if i > 0:  # the previous search's times: waits for that search only, this one is already queued
            corr_ms.append(eng.last_timing(1)["ms_correlate"])
    corr_ms.append(eng.last_timing(0)["ms_correlate"])
    fence()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    timing = eng.last_timing()

    if rank == 0:
        total_cells = job_cells_per_step * args.steps
        value = total_cells / elapsed
        kern_ms = float(np.mean(corr_ms))
        achieved = cells_per_step * ALG_BYTES_PER_CELL / (kern_ms * 1e-3) / 1e9
        traffic, traffic_src, onchip = None, None, None
        try:  # HBM bytes per launch from the committed PMC passes (profiles/traffic.json), scaled by cells
            tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
            traffic = (tj["hbm_read_bytes_per_cell"] + tj["hbm_write_bytes_per_cell"]) * cells_per_step
            traffic_src = f"profiles/{tj['tag']}_summary.md (rocprofv3 --pmc FETCH_SIZE, WRITE_SIZE; bytes per cell x cells per launch)"
            onchip = tj.get("onchip_counters")
        except Exception:
            pass
        out = {
            "metric": "(PRN,Doppler) correlation cells/s, 32 PRN @ fs=5.456 MHz",
            "value": value,
            "unit": "cells/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "strong" if grid else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "data_detail": ("device-generated 1-bit real-IF capture: white noise + PRNs %s at 45 dB-Hz, seeded Doppler/code phase" % injected)
                           if args.data == "signals" else "uniform random bits (sign of white noise)",
            "config": {"workload": (f"BASELINE configs[4]: {args.grid_blocks} blocks x 32 PRN x {2 * eng.dmax + 1} Doppler bins (+-100 kHz, "
                                    "fs/N = 136.4 Hz), N=40000, 5456 lags") if grid else
                                   ("BASELINE configs[1]: 32 PRN x 73 Doppler bins (+-5 kHz, fs/N = 136.4 Hz), N=40000, "
                                    "5456 lags, reference schedule block->PRN (block % 32)"),
                       "fs_hz": FS, "if_hz": FC, "blocks_per_gpu": nblk, "cells_per_step_per_gpu": cells_per_step,
                       "parallelism": (f"Doppler slabs sharded over {world} GPU(s), per-(block, PRN) peak all-reduce(MAX)" if grid else
                                       f"blocks sharded over {world} GPU(s), per-PRN peak all-reduce(MAX)")},
            "roofline": {"bound": "hbm", "kernel": f"k_corr<{eng.acc_columns}>", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": cells_per_step * ALG_BYTES_PER_CELL,
                         "note": "fused kernel: the algorithmic bytes never reach HBM (traffic << algorithmic), so frac > 1; "
                                 "the real limiters are the fp32 VALU and LDS pipes (DESIGN.md section 4)",
                         "onchip_counters": onchip,
                         "kernel_ms": kern_ms, "algorithmic_bytes_per_cell": ALG_BYTES_PER_CELL,
                         "cells_per_launch": cells_per_step},
            "stage_ms": {k: timing[k] for k in ("ms_total", "ms_sample", "ms_correlate", "ms_peaks")},
            "device": eng.device_name,
        }
        if args.data == "signals" and not grid:  # the search must actually see what was injected (rank 0's capture)
            snr, lo, ca = gdist.unpack_keys(best.cpu(), eng.dmax)
            out["detected_prns"] = [int(p) + 1 for p in torch.nonzero(snr >= 25).flatten().tolist()]
            out["injected_prns_rank0"] = injected
            # after the all-reduce the per-PRN best covers every rank's capture (rank r: seed 1000 + r)
            out["injected_prns_all_ranks"] = sorted(set().union(*[synth_sats(1000 + r)[0] for r in range(world)]))
        if world == 1 and not args.no_cpu_baseline:
            out["roofline"]["hbm_copy_measured_GBs"] = hbm_copy_gbs(torch, dev)
            out["cpu_baseline"] = cpu_baseline(host_bits)
            try:
                out["cpu_baseline_all_cores"] = cpu_baseline_all_cores(host_bits, out["cpu_baseline"]["value"])
            except Exception as ex:  # the 1-thread figure is the contract; this one is informative
                out["cpu_baseline_all_cores"] = {"error": str(ex)}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
