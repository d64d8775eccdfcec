#!/usr/bin/env python3
"""tools/write_ceiling.py -- what a write-only stream reaches on this GPU (k_fwd writes 320 KB of spectrum per 5 KB block read:
its roofline is the HBM *write* rate, not the read+write copy ceiling).  torch fill_ / zero_ / hipMemsetAsync of 3.5 GB, the bytes
k_fwd writes per Nottingham-size launch; also a read-only reduction for the other side."""
import json
import torch

dev = torch.device("cuda", 0)
n = 3_481_600_000 // 4
x = torch.empty(n, dtype=torch.float32, device=dev)
out = {}


def timed(name, fn, nbytes, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / reps
    out[name] = {"ms": ms, "GBs": nbytes / ms / 1e6}


timed("fill_f32", lambda: x.fill_(1.5), n * 4)
timed("zero_", lambda: x.zero_(), n * 4)
y = torch.empty_like(x)
timed("copy_ (read + write)", lambda: y.copy_(x), 2 * n * 4)
timed("sum (read only)", lambda: x.sum(), n * 4)
x16 = x.view(torch.float64)
timed("fill_f64", lambda: x16.fill_(2.5), n * 4)
print(json.dumps(out, indent=1))
