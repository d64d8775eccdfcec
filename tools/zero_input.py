#!/usr/bin/env python3
"""Power-limit check (needs an MI355X): the correlate kernel on random capture bits against all-zero bits -- the same
instruction stream on data that toggles far fewer multiplier inputs.  A faster zero run means the chip clock is set by
power, not by the kernel.  Run from the repo root."""
import os, sys
sys.path.insert(0, "gnss-gps-sdr_amd/python")


def main():
    import torch, gpsacq
    with gpsacq.Engine(4.092e6, 5.456e6, 5000.0) as eng:
        nblk = 4096
        for name, d_bits in (("random bits", torch.randint(0, 256, (nblk * 5120,), dtype=torch.uint8, device="cuda")),
                             ("all-zero bits", torch.zeros(nblk * 5120, dtype=torch.uint8, device="cuda")),
                             ("random bits again", torch.randint(0, 256, (nblk * 5120,), dtype=torch.uint8, device="cuda"))):
            d_peaks = torch.zeros(nblk * 4, dtype=torch.int32, device="cuda")
            for _ in range(3):
                eng.search_device(d_bits.data_ptr(), nblk, d_peaks.data_ptr())
            ms = []
            for _ in range(8):
                eng.search_device(d_bits.data_ptr(), nblk, d_peaks.data_ptr())
                ms.append(eng.last_timing()["ms_correlate"])
            k = sum(ms) / len(ms)
            print(f"{name:18s} k_corr {k:.3f} ms = {nblk * 73 / k / 1e3:.2f} M cells/s")


if __name__ == "__main__":
    main()
