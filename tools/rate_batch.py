#!/usr/bin/env python3
"""In-kernel and whole-search rate against the batch size (blocks per launch) at BASELINE configs[1]: what strong scaling over
N GPUs does to each GPU's share (10 880 blocks / N).  tools/rate_batch.py  (needs an MI355X)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gnss-gps-sdr_amd", "python"))
import torch  # noqa: E402
import gpsacq  # noqa: E402

with gpsacq.Engine(4.092e6, 5.456e6, 5000.0) as eng:
    for nblk in (64, 256, 512, 1024, 1376, 2720, 5440, 10880):
        d_bits = torch.randint(0, 256, (nblk * 5120,), dtype=torch.uint8, device="cuda")
        d_peaks = torch.zeros(nblk * 4, dtype=torch.int32, device="cuda")
        for _ in range(3):
            eng.search_device(d_bits.data_ptr(), nblk, d_peaks.data_ptr())
        ms, tot = [], []
        for _ in range(6):
            eng.search_device(d_bits.data_ptr(), nblk, d_peaks.data_ptr())
            t = eng.last_timing()
            ms.append(t["ms_correlate"])
            tot.append(t["ms_total"])
        cells = nblk * eng.num_doppler
        k, w = sum(ms) / len(ms), sum(tot) / len(tot)
        print(f"blocks {nblk:6d}  cells {cells:7d}  k_corr {k:8.3f} ms = {cells / k / 1e3:6.2f} M cells/s   whole search {w:8.3f} ms = {cells / w / 1e3:6.2f} M cells/s")
