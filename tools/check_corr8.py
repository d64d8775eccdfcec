#!/usr/bin/env python3
"""The 8-wave correlator experiment (k_corr8, acq_corr8.hpp: 5 x 10 x 10 x 10 on 500 threads) against the product's k_corr, on
the GPU box.  k_corr8 is NOT in libgpsacq.so: `make experiments` builds build/var_exp/libgpsacq.so with it.  Cells of the three
fixtures: powers to 2e-6, identical argmax (the round-3 product test, moved here with the kernel).
Usage: make experiments && python tools/check_corr8.py"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXP = os.path.join(ROOT, "build", "var_exp", "libgpsacq.so")
assert os.path.exists(EXP), "run `make experiments` first"
child = ("import sys, numpy as np; sys.path.insert(0, %r); import gpsacq\n"
         "buf = open(sys.argv[2], 'rb').read()[:33 * 5120]\n"
         "with gpsacq.Engine(float(sys.argv[3]), float(sys.argv[4]), 5000.0) as e:\n"
         "    c, p = e.search(buf)\n"
         "np.save(sys.argv[1], c)\n" % os.path.join(ROOT, "gnss-gps-sdr_amd", "python"))
for fc, fs, name in ((4.092e6, 5.456e6, "synth_nott_fs5456.bin"), (2.046e6, 8.184e6, "gps_sig_tmp.bin"), (0.62e6, 2.8e6, "synth_rtl_fs2800.bin")):
    with tempfile.TemporaryDirectory() as d:
        outs = []
        for env in ({}, {"GPSACQ_LIB": EXP, "GPSACQ_CORR8": "2"}):
            f = os.path.join(d, "c%d.npy" % len(outs))
            subprocess.run([sys.executable, "-c", child, f, os.path.join(ROOT, "tests", "golden", name), repr(fc), repr(fs)],
                           env=dict(os.environ, **env), check=True, timeout=300)
            outs.append(np.load(f))
    a, b = outs
    assert not np.array_equal(a["tot_pwr"], b["tot_pwr"])  # it really was another kernel
    np.testing.assert_allclose(b["max_pwr"], a["max_pwr"], rtol=2e-6)
    np.testing.assert_allclose(b["tot_pwr"], a["tot_pwr"], rtol=2e-6)
    assert (a["max_i"] != b["max_i"]).sum() <= 1
    print(name, "k_corr8 == k_corr: max_pwr rel", float(np.max(np.abs(b["max_pwr"] / a["max_pwr"] - 1))))
