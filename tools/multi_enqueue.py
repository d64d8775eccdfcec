#!/usr/bin/env python3
"""gpsacq_multi_search_blocks on a capture of the Nottingham size (340 runs = 10 880 blocks, 55.7 MB of pageable host memory)
with 1, 2, 4, 8 engines sharing the one GPU: host time until everything is enqueued (gpsacq_multi_last_call_ms) next to the
whole call.  With GPSACQ_MULTI_FORCE_RCCL=1 the merge goes through the one-rank RCCL communicator.  Prints one JSON line.
Usage: python tools/multi_enqueue.py [n_runs]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gnss-gps-sdr_amd", "python"))
import gpsacq  # noqa: E402

n_runs = int(sys.argv[1]) if len(sys.argv) > 1 else 340
fc, fs = 4.092e6, 5.456e6
with gpsacq.Engine(fc, fs, 5000.0) as eng:
    sats = [(3, 0.151, 1200.0, 100.0, 0.1), (17, 0.151, -2300.0, 2500.0, 0.3), (28, 0.151, 400.0, 4000.0, 0.7)]
    buf = eng.generate(n_runs * 32 * 5120, sats, noise_sigma=1.0, seed=11)
    t0 = time.perf_counter()
    _, want = eng.search(buf, want_cells=False)
    plain_ms = 1e3 * (time.perf_counter() - t0)
out = {"capture_runs": n_runs, "capture_bytes": int(buf.size), "plain_gpsacq_search_ms_first_call": plain_ms,
       "force_rccl": os.environ.get("GPSACQ_MULTI_FORCE_RCCL", "0"), "rows": []}
for n in (1, 2, 4, 8):
    with gpsacq.MultiEngine(fc, fs, 5000.0, devices=(0,) * n) as me:
        me.search_blocks(buf)
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            peaks, best = me.search_blocks(buf)
            wall = 1e3 * (time.perf_counter() - t0)
            t = me.last_call_ms()
            t["wall_ms"] = wall
            ts.append(t)
        assert np.array_equal(peaks, want)
        out["rows"].append({"engines": n, "enqueue_ms_min": min(t["enqueue_ms"] for t in ts), "enqueue_ms_median": float(np.median([t["enqueue_ms"] for t in ts])),
                            "total_ms_min": min(t["total_ms"] for t in ts), "wall_ms_min": min(t["wall_ms"] for t in ts),
                            "rccl_allreduces": ts[-1]["rccl_allreduces"], "peaks_equal_plain_search": True})
print(json.dumps(out))
