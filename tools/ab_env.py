#!/usr/bin/env python3
"""A/B of an environment-selected kernel instance against the default one on the GPU box: cells compared bit for bit and
the in-kernel rate timed in separate processes.  tools/ab_env.py VAR=1 [VAR2=1 ...]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.join(%r, "gnss-gps-sdr_amd", "python"))
import torch, gpsacq
buf = open(os.path.join(%r, "tests", "golden", "synth_nott_fs5456.bin"), "rb").read()[:40 * 5120]
out = {}
with gpsacq.Engine(4.092e6, 5.456e6, 5000.0) as eng:
    cells, peaks = eng.search(buf)
    if os.environ.get("AB_SAVE"):
        np.save(os.environ["AB_SAVE"], cells)
    else:
        r = np.load(os.environ["AB_REF"])
        out["bit_exact"] = bool(np.array_equal(cells, r))
        out["max_rel_pwr"] = float(np.max(np.abs(cells["max_pwr"] / r["max_pwr"] - 1)))
        out["argmax_mismatch"] = int((cells["max_i"] != r["max_i"]).sum())
    nblk = 4096
    d_bits = torch.randint(0, 256, (nblk * 5120,), dtype=torch.uint8, device="cuda")
    d_peaks = torch.zeros(nblk * 4, dtype=torch.int32, device="cuda")
    for _ in range(2):
        eng.search_device(d_bits.data_ptr(), nblk, d_peaks.data_ptr())
    ms = []
    for _ in range(8):
        eng.search_device(d_bits.data_ptr(), nblk, d_peaks.data_ptr())
        ms.append(eng.last_timing()["ms_correlate"])
    out["ms_fwd"] = eng.last_timing()["ms_sample"]
    out["ms_avg"] = sum(ms) / len(ms)
    out["ms_min"] = min(ms)
    out["mcells_s"] = nblk * eng.num_doppler / out["ms_avg"] / 1e3
print("ABOUT " + json.dumps(out))
''' % (ROOT, ROOT)


def run(extra):
    env = dict(os.environ)
    env.update(extra)
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=900)
    for line in r.stdout.splitlines():
        if line.startswith("ABOUT "):
            return json.loads(line[6:])
    return {"error": (r.stdout + r.stderr)[-600:]}


ref = "/tmp/ab_ref.npy"
print("default      ", json.dumps(run({"AB_SAVE": ref})), flush=True)
for arg in sys.argv[1:]:
    k, _, v = arg.partition("=")
    print("%-12s " % arg, json.dumps(run({k: v or "1", "AB_REF": ref})), flush=True)
print("default again", json.dumps(run({"AB_REF": ref})), flush=True)
