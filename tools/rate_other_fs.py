#!/usr/bin/env python3
"""In-kernel correlate rate (cells/s) at other sampling rates / accumulator-column instances than the
bench's: tools/rate_other_fs.py  (needs an MI355X).  Prints one line per configuration."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gnss-gps-sdr_amd", "python"))
import torch  # noqa: E402
import gpsacq  # noqa: E402

for fc, fs, max_fo, nblk in [(4.092e6, 5.456e6, 5000.0, 2048), (0.62e6, 2.8e6, 5000.0, 1024), (1.7e6, 6.8e6, 5000.0, 2048), (2.046e6, 8.184e6, 5000.0, 2048),
                             (2.6e6, 10e6, 5000.0, 2048), (3.0e6, 12e6, 5000.0, 1024), (4.0e6, 16.368e6, 5000.0, 1024)]:
    with gpsacq.Engine(fc, fs, max_fo) as eng:
        d_bits = torch.randint(0, 256, (nblk * 5120,), dtype=torch.uint8, device="cuda")
        d_peaks = torch.zeros(nblk * 4, dtype=torch.int32, device="cuda")
        for _ in range(2):
            eng.search_device(d_bits.data_ptr(), nblk, d_peaks.data_ptr())
        ms = []
        for _ in range(4):
            eng.search_device(d_bits.data_ptr(), nblk, d_peaks.data_ptr())
            ms.append(eng.last_timing()["ms_correlate"])
        t = eng.last_timing()
        cells = nblk * eng.num_doppler
        k = sum(ms) / len(ms)
        print(f"fs {fs/1e6:7.3f} MHz  lags {eng.num_lags:5d}  columns {eng.acc_columns:2d}  bins {eng.num_doppler:3d}  "
              f"launches {t['correlate_launches']}  {cells} cells in {k:.2f} ms = {cells / k / 1e3:.2f} M cells/s in-kernel")
