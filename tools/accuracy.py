#!/usr/bin/env python3
"""Measured numerical agreement of the HIP path with the float64 oracle (for DESIGN.md)."""
import os, sys
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(root, "gnss-gps-sdr_amd", "python")); sys.path.insert(0, os.path.join(root, "tests"))
import gpsacq
from oracle_lib import Oracle
for name, fc, fs, f in [("nott", 4.092e6, 5.456e6, "synth_nott_fs5456.bin"), ("sigtmp", 2.046e6, 8.184e6, "gps_sig_tmp.bin"), ("rtl", 0.62e6, 2.8e6, "synth_rtl_fs2800.bin")]:
    buf = open(os.path.join(root, "tests", "golden", f), "rb").read()[:33 * 5120]
    sel = list(range(0, 33, 2))
    with gpsacq.Engine(fc, fs, 5000.0) as eng:
        cells, peaks = eng.search(buf)
    oc, op = Oracle(fc, fs, 5000.0).search(buf, [(b, b % 32) for b in sel])
    g = cells[sel]
    e1 = np.abs(g["max_pwr"] / oc["max_pwr"] - 1); e2 = np.abs(g["tot_pwr"] / oc["tot_pwr"] - 1); e3 = np.abs(g["snr"] / oc["snr"] - 1)
    print(f"{name}: cells {g.size}  max_pwr rel err max {e1.max():.2e} median {np.median(e1):.2e} | tot_pwr max {e2.max():.2e} | snr max {e3.max():.2e} | argmax mismatches {(g['max_i'] != oc['max_i']).sum()}")
