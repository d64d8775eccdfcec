#!/usr/bin/env python3
"""k_corr<22>'s rate against the workgroups resident per CU (1, 2, 3): unused dynamic LDS keeps the others out
(GPSACQ_CORR_LDS_PAD, experiment library only).  From the three points a closed-system reading: throughput X(n) = n / (D (n - 1) q + R1)
... printed raw; profiles/HISTORY.md interprets.  Usage: make experiments && python tools/residency_curve.py"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXP = os.path.join(ROOT, "build", "var_exp", "libgpsacq.so")
CHILD = r'''
import os, sys, json
sys.path.insert(0, os.path.join(%r, "gnss-gps-sdr_amd", "python"))
import torch, gpsacq
with gpsacq.Engine(4.092e6, 5.456e6, 5000.0) as eng:
    nblk = 2048
    d_bits = torch.randint(0, 256, (nblk * 5120,), dtype=torch.uint8, device="cuda")
    d_peaks = torch.zeros(nblk * 4, dtype=torch.int32, device="cuda")
    for _ in range(2):
        eng.search_device(d_bits.data_ptr(), nblk, d_peaks.data_ptr())
    ms = []
    for _ in range(6):
        eng.search_device(d_bits.data_ptr(), nblk, d_peaks.data_ptr())
        ms.append(eng.last_timing()["ms_correlate"])
    print("OUT " + json.dumps({"ms_avg": sum(ms) / len(ms), "mcells_s": nblk * eng.num_doppler / (sum(ms) / len(ms)) / 1e3}))
''' % ROOT
rows = []
for wgs, pad in ((3, 0), (2, 31000), (1, 60000), (3, 0)):
    env = dict(os.environ, GPSACQ_LIB=EXP)
    if pad:
        env["GPSACQ_CORR_LDS_PAD"] = str(pad)
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=600)
    out = [l for l in r.stdout.splitlines() if l.startswith("OUT ")]
    rows.append({"workgroups_per_cu": wgs, "lds_pad": pad, **(json.loads(out[0][4:]) if out else {"error": (r.stdout + r.stderr)[-300:]})})
    print(json.dumps(rows[-1]), flush=True)
