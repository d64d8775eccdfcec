#!/usr/bin/env python3
"""Wall-clock of the gps_test front end on a Nottingham-size synthetic capture (55,791,616 bytes =
10,896 blocks, 340 complete runs, 794,240 cells): file read + PCIe + search + report."""
import os, subprocess, sys, time
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = "/tmp/nott_size.bin"
np.random.default_rng(0).integers(0, 256, 446332928 // 8, dtype=np.uint8).tofile(path)
exe = os.path.join(root, "gnss-gps-sdr_amd", "bin", "gps_test")
for batch in ("64", "340"):
    for rep in range(2):
        t0 = time.perf_counter()
        out = subprocess.run([exe, path, "4.092e6", "5.456e6", "5000"], capture_output=True, text=True,
                             env=dict(os.environ, GPSACQ_BATCH_RUNS=batch)).stdout
        dt = time.perf_counter() - t0
        print(f"batch_runs={batch} rep={rep}: wall {dt:.3f} s, runs {out.count('satellite:')}, {340*32*73/dt/1e6:.2f} M cells/s end to end")
