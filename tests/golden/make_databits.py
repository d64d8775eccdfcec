"""Recovers the 100 navigation bits of the reference's bundled gps_sig_tmp.bin (gps_sig_gen.m draws them with an unseeded
rand, :18) and stores them as tests/golden/gps_sig_tmp_databits.json.  Run from the repo root: python tests/golden/make_databits.py"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
from oracle_lib import lib, _p  # noqa: E402
import sig_gen_oracle as sg  # noqa: E402

chips = np.zeros(1023, np.uint8)
lib("f64").oracle_ca_chips(7, _p(chips))  # sv index 7 = PRN 8 (gps_sig_gen.m:11)
raw = open(os.path.join(HERE, "gps_sig_tmp.bin"), "rb").read()
data, worst = sg.recover_data_bits(raw, chips)
assert worst == 1.0, worst
json.dump({"source": "recovered from tests/golden/gps_sig_tmp.bin (= the reference's gps_sig_tmp.bin) by tests/golden/make_databits.py",
           "prn": 8, "bits_pm1": [int(v) for v in data]}, open(os.path.join(HERE, "gps_sig_tmp_databits.json"), "w"))
print(len(data), "bits, consistency", worst)
