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
                      f"{dt:.1f} s on {os.cpu_count()} core host, 1 thread"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--blocks", type=int, default=2048, help="5120-byte blocks per GPU per step (64 runs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
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

    eng = gpsacq.Engine(FC, FS, MAX_FO, device=dev_index)
    nblk = args.blocks
    host_bits = synth_bits(nblk, 1000 + rank)
    dev = torch.device("cuda", dev_index)
    d_bits = torch.from_numpy(host_bits).to(dev)
    d_peaks = torch.zeros((nblk, 4), dtype=torch.int32, device=dev)
    cells_per_step = nblk * eng.num_doppler

    def step():
        eng.search_device(d_bits.data_ptr(), nblk, d_peaks.data_ptr(), sync=True)
        # per-PRN best peak of this rank's capture, packed so that integer MAX reproduces the
        # reference's ordering (higher SNR; ties -> lower Doppler bin, :198)
        snr_bits = d_peaks[:, 0].to(torch.int64)
        lo = d_peaks[:, 1].to(torch.int64) + eng.dmax
        ca = d_peaks[:, 2].to(torch.int64)
        key = (snr_bits << 32) | ((0xFFFF - lo) << 16) | ca
        best = key.view(-1, 32).max(dim=0).values
        if dist is not None:
            if backend == "nccl":
                dist.all_reduce(best, op=dist.ReduceOp.MAX)  # RCCL over xGMI, 256 bytes
            else:
                b = best.cpu()
                dist.all_reduce(b, op=dist.ReduceOp.MAX)
                best = b.to(dev)
        return best

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    corr_ms = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        best = step()
        corr_ms.append(eng.last_timing()["ms_correlate"])
    fence()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    timing = eng.last_timing()

    if rank == 0:
        total_cells = cells_per_step * world * args.steps
        value = total_cells / elapsed
        kern_ms = float(np.mean(corr_ms))
        achieved = cells_per_step * ALG_BYTES_PER_CELL / (kern_ms * 1e-3) / 1e9
        traffic, traffic_src = None, None
        try:  # HBM bytes per launch from the committed PMC passes (profiles/traffic.json), scaled by cells
            tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
            traffic = (tj["hbm_read_bytes_per_cell"] + tj["hbm_write_bytes_per_cell"]) * cells_per_step
            traffic_src = f"profiles/{tj['tag']}_summary.md (rocprofv3 --pmc FETCH_SIZE, WRITE_SIZE; bytes per cell x cells per launch)"
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
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: 32 PRN x 73 Doppler bins (+-5 kHz, fs/N = 136.4 Hz), N=40000, "
                                   "5456 lags, reference schedule block->PRN (block % 32)",
                       "fs_hz": FS, "if_hz": FC, "blocks_per_gpu": nblk, "cells_per_step_per_gpu": cells_per_step,
                       "parallelism": f"blocks sharded over {world} GPU(s), per-PRN peak all-reduce(MAX)"},
            "roofline": {"bound": "hbm", "kernel": f"k_corr<{eng.acc_columns}>", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": cells_per_step * ALG_BYTES_PER_CELL,
                         "kernel_ms": kern_ms, "algorithmic_bytes_per_cell": ALG_BYTES_PER_CELL,
                         "cells_per_launch": cells_per_step},
            "stage_ms": {k: timing[k] for k in ("ms_total", "ms_sample", "ms_correlate", "ms_peaks")},
            "device": eng.device_name,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(host_bits)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
