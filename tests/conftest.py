import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gnss-gps-sdr_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
GOLDEN = os.path.join(ROOT, "tests", "golden")

ARTIFACTS = [
    os.path.join(ROOT, "gnss-gps-sdr_amd", "lib", "libgpsacq.so"),
    os.path.join(ROOT, "gnss-gps-sdr_amd", "lib", "libgps_search.so"),
    os.path.join(ROOT, "gnss-gps-sdr_amd", "bin", "gps_test"),
    os.path.join(ROOT, "oracle", "liboracle_f64.so"),
    os.path.join(ROOT, "oracle", "liboracle_f32.so"),
    os.path.join(ROOT, "tests", "emul", "libemul_acq.so"),
]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # Built artefacts are git-ignored: build them once if a fresh checkout lacks any
    # (same recipe as __graft_entry__.build(); hipcc cross-compiles gfx950 without a GPU).
    if not all(os.path.exists(p) for p in ARTIFACTS):
        subprocess.check_call(["make", "-C", ROOT, "lib", "host", "oracle", "emul"], stdout=subprocess.DEVNULL)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
