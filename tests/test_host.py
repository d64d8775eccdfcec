"""Host side: the gps_test front end and the SearchTask report format (no GPU needed)."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GPS_TEST = os.path.join(ROOT, "gnss-gps-sdr_amd", "bin", "gps_test")
BANNER = ("GPS CA code offline search. Extract from http://www.aholme.co.uk/GPS/Main.htm\n"
          "Jiao Xianjun (putaoshu@gmail.com). 2014-05.\n"
          "usage:\n"
          "gps_test   filename_of_1bit_IF_cap   carrier_freq   sampling_rate   max_freq_offset\n"
          "or\n"
          "gps_test (Make sure gps.samples.1bit.I.fs5456.if4092.bin can be found. Download http://www.jks.com/gps/gps.html)\n")


def _ensure_built():
    pass  # the hip_artifacts fixture (conftest.py) built the front end


@pytest.mark.usefixtures("hip_artifacts")
def test_cli_banner_and_arg_count():
    """c/test_search_offline.cpp:24-38: six banner lines always; argc not in {1,5} -> message, exit 0."""
    _ensure_built()
    r = subprocess.run([GPS_TEST, "a", "b"], capture_output=True, text=True)
    assert r.returncode == 0
    assert r.stdout == BANNER + "Please run with 3 arguments or without argument!\n"


@pytest.mark.usefixtures("hip_artifacts")
def test_cli_without_gpu_reports_init_failure():
    import torch
    if torch.cuda.is_available():
        return
    _ensure_built()
    r = subprocess.run([GPS_TEST, "x.bin", "4.092e6", "5.456e6", "5000"], capture_output=True, text=True)
    assert r.stdout == BANNER + "SearchInit() returned 2\n" and r.returncode == 2
    assert "no CPU path" in r.stderr


def test_report_format_matches_oracle(golden_dir):
    """gpsacq.format_report() (the Python twin of SearchTask's printf block) reproduces the oracle's
    SearchTask text when fed the oracle's own peaks."""
    import gpsacq
    from oracle_lib import Oracle
    orc = Oracle(4.092e6, 5.456e6, 5000.0)
    n, text, peaks = orc.search_file(os.path.join(golden_dir, "synth_nott_fs5456.bin"), max_runs=1)
    assert n == 1
    assert gpsacq.format_report(peaks) == text
    assert " 0 satellite:     0    20    28    29    30 \n" in text
