import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gnss-gps-sdr_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
GOLDEN = os.path.join(ROOT, "tests", "golden")

# Built artefacts are git-ignored.  The CPU suite needs only the oracle (gcc) and the emulation harness (clang);
# the HIP library, the host front end and everything marked `gpu` need hipcc and are built on first use.
CPU_ARTIFACTS = [
    os.path.join(ROOT, "oracle", "liboracle_f64.so"),
    os.path.join(ROOT, "oracle", "liboracle_f32.so"),
    os.path.join(ROOT, "tests", "emul", "libemul_acq.so"),
]
HIP_ARTIFACTS = [
    os.path.join(ROOT, "gnss-gps-sdr_amd", "lib", "libgpsacq.so"),
    os.path.join(ROOT, "gnss-gps-sdr_amd", "lib", "libgps_search.so"),
    os.path.join(ROOT, "gnss-gps-sdr_amd", "bin", "gps_test"),
]


def _make(*targets):
    r = subprocess.run(["make", "-C", ROOT, *targets], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("make %s failed:\n%s\n%s" % (" ".join(targets), r.stdout[-4000:], r.stderr[-4000:]))


def build_hip_artifacts():
    """libgpsacq.so / libgps_search.so / gps_test (same recipe as __graft_entry__.build()); hipcc cross-compiles
    gfx950 without a GPU.  Returns a reason string if they cannot be built here."""
    if all(os.path.exists(p) for p in HIP_ARTIFACTS):
        return None
    if not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        return "hipcc not found: the HIP library cannot be built on this machine"
    _make("lib", "host")
    return None


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    if not all(os.path.exists(p) for p in CPU_ARTIFACTS):
        _make("oracle", "emul")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def hip_artifacts():
    """Tests that load libgpsacq.so or run the front end (also the CPU-side ABI / fail-loud tests) depend on this."""
    why = build_hip_artifacts()
    if why:
        pytest.skip(why)
    return HIP_ARTIFACTS


@pytest.fixture(autouse=True)
def _hip_artifacts_for_gpu_tests(request):
    """gpu-marked tests always need the HIP artefacts (built on first use; on the GPU box they arrive prebuilt)"""
    if request.node.get_closest_marker("gpu"):
        request.getfixturevalue("hip_artifacts")
