#!/usr/bin/env python3
"""Does the forward transform hide under the correlator?  Two engines (two HIP streams) on one GPU: A runs full searches
(k_corr-bound), B runs searches of ONE Doppler bin (k_fwd-bound: the forward transform is 2/3 of such a search).  Timed alone and
together; overlap gain = T_A + T_B - T_both.  (needs an MI355X; python tools/overlap_probe.py)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gnss-gps-sdr_amd", "python"))
import torch, gpsacq
nblk = 4096
with gpsacq.Engine(4.092e6, 5.456e6, 5000.0) as A, gpsacq.Engine(4.092e6, 5.456e6, 5000.0) as B:
    B.set_doppler_window(0, 1)
    bits = torch.randint(0, 256, (nblk * 5120,), dtype=torch.uint8, device="cuda")
    pa = torch.zeros(nblk * 4, dtype=torch.int32, device="cuda")
    pb = torch.zeros(nblk * 4, dtype=torch.int32, device="cuda")
    def run(na, nb):
        torch.cuda.synchronize(); A.synchronize(); B.synchronize()
        t0 = time.perf_counter()
        ia = ib = 0
        while ia < na or ib < nb:  # interleave the submissions in proportion
            if ia < na and (ib >= nb or ia * nb <= ib * na):
                A.search_device(bits.data_ptr(), nblk, pa.data_ptr(), sync=False); ia += 1
            else:
                B.search_device(bits.data_ptr(), nblk, pb.data_ptr(), sync=False); ib += 1
        A.synchronize(); B.synchronize()
        return (time.perf_counter() - t0) * 1e3
    run(2, 8)
    na, nb = 8, 96
    for rep in range(2):
        ta, tb, tboth = run(na, 0), run(0, nb), run(na, nb)
        tm = B.last_timing()
        print(f"A alone {ta:.2f} ms ({na} searches)  B alone {tb:.2f} ms ({nb} one-bin searches; last: fwd {tm['ms_sample']:.3f} corr {tm['ms_correlate']:.3f})  "
              f"both {tboth:.2f} ms  -> hidden {ta + tb - tboth:.2f} ms of B's {tb:.2f}")
