#!/usr/bin/env python3
"""Front-end wall clock on a capture file of the Nottingham size (340 runs, 55.7 MB in /dev/shm), alternating plain runs with
GPSACQ_WARM=1 runs (the first launch of each search kernel made by a worker right after gpsacq_create, while SearchTask
allocates its staging buffers): wall time and the front end's own GPSACQ_TRACE split.  Prints one JSON line."""
import json
import os
import re
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gnss-gps-sdr_amd", "python"))
import gpsacq  # noqa: E402

n_runs, reps = 340, int(sys.argv[1]) if len(sys.argv) > 1 else 7
path = "/dev/shm/gpsacq_e2e_warm.bin" if os.path.isdir("/dev/shm") else "/tmp/gpsacq_e2e_warm.bin"
with gpsacq.Engine(4.092e6, 5.456e6, 5000.0) as eng:
    eng.generate(n_runs * 32 * 5120, [(3, 0.151, 1200.0, 100.0, 0.1), (17, 0.151, -2300.0, 2500.0, 0.3)], noise_sigma=1.0, seed=5).tofile(path)
exe = os.path.join(ROOT, "gnss-gps-sdr_amd", "bin", "gps_test")
out = {"file_bytes": os.path.getsize(path), "runs": n_runs, "modes": {}}
ref = None
for rep in range(reps):
    for mode, env in (("plain", {}), ("warm", {"GPSACQ_WARM": "1"})):
        t0 = time.perf_counter()
        r = subprocess.run([exe, path, "4.092e6", "5.456e6", "5000"], capture_output=True, text=True, env=dict(os.environ, GPSACQ_TRACE="1", **env))
        wall = time.perf_counter() - t0
        assert r.returncode == 0, r.stderr[-500:]
        if ref is None:
            ref = r.stdout
        assert r.stdout == ref  # the same report either way
        tr = [ln for ln in r.stderr.splitlines() if ln.startswith("gpsacq trace: SearchInit")][-1]
        nums = {k: float(v) for k, v in re.findall(r"(SearchInit|SearchTask|buffers|read|submit|wait for GPU|report) ([0-9.]+)", tr)}
        nums["wall_ms"] = 1e3 * wall
        out["modes"].setdefault(mode, []).append(nums)
for mode, rows in out["modes"].items():
    out[mode + "_median"] = {k: float(np.median([r[k] for r in rows])) for k in rows[0]}
    out[mode + "_min"] = {k: float(np.min([r[k] for r in rows])) for k in rows[0]}
del out["modes"]
os.remove(path)
print(json.dumps(out))
