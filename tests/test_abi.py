"""The C-ABI library loads and exports every symbol include/gpsacq.h declares; without a GPU
the product fails loudly (no CPU fallback).  No compute calls here."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.usefixtures("hip_artifacts")  # built on first use (conftest.py); skipped where hipcc is missing


def _header_symbols(path):
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gpsacq_[a-z0-9_]+)\s*\(", text)))


def test_exports_match_header():
    import gpsacq
    lib = gpsacq.load_library()
    syms = _header_symbols(os.path.join(ROOT, "include", "gpsacq.h"))
    assert len(syms) >= 12
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/gpsacq.h but not exported"
    assert sorted(gpsacq.EXPORTS) == syms


def test_gps_search_library_exports_the_reference_api():
    """include/gps_search.h: the reference's five entry points (c/gps_offline.h:87-91, C++ linkage, so mangled names) plus
    SearchStatus(); the three globals FC / FS / max_fo stay undefined in the library -- the caller defines them
    (c/test_search_offline.cpp:12)."""
    lib = os.path.join(ROOT, "gnss-gps-sdr_amd", "lib", "libgps_search.so")
    out = subprocess.run(["nm", "-D", "-C", lib], capture_output=True, text=True, check=True).stdout
    defined = {l.split(" T ", 1)[1].strip() for l in out.splitlines() if " T " in l}
    for sym in ("SearchInit()", "SearchFree()", "SearchTask(char*)", "SearchEnable(int)", "SearchCode(int, int)", "SearchStatus()"):
        assert sym in defined, sym
    undefined = {l.split(" U ", 1)[1].strip() for l in out.splitlines() if " U " in l}
    assert {"FC", "FS", "max_fo"} <= undefined
    header = open(os.path.join(ROOT, "include", "gps_search.h")).read()
    for name in ("SearchInit", "SearchFree", "SearchTask", "SearchEnable", "SearchCode", "SearchStatus", "FC", "FS", "max_fo"):
        assert re.search(r"\b%s\b" % name, header)


def test_struct_sizes_match_abi():
    import gpsacq
    assert gpsacq.CELL_DTYPE.itemsize == 16 and gpsacq.PEAK_DTYPE.itemsize == 16 and gpsacq.TASK_DTYPE.itemsize == 8
    assert ctypes.sizeof(gpsacq.Params) == 32 and ctypes.sizeof(gpsacq.Info) == 120 and ctypes.sizeof(gpsacq.Timing) == 32


def test_search_code_host_only():
    """SearchCode() (c/search_offline.cpp:205-209) is host arithmetic: check it against the oracle."""
    import gpsacq
    from oracle_lib import lib
    L = lib("f64")
    for sv in (0, 7, 31):
        for g1 in (0x3FF, 0x1, 0x2AA, 0x155, 0x3FE):
            assert gpsacq.search_code(sv, g1) == L.oracle_search_code(sv, g1)
    assert gpsacq.search_code(40, 1) == -1


def test_argument_errors_before_device():
    import gpsacq
    with pytest.raises(gpsacq.GpsAcqError) as ei:
        gpsacq.Engine(1e6, 0.0)
    assert ei.value.code == 1
    with pytest.raises(gpsacq.GpsAcqError) as ei:
        gpsacq.Engine(1e6, 5e6, 3e6)  # Doppler range beyond half the sampling rate
    assert ei.value.code == 3
    with pytest.raises(gpsacq.GpsAcqError) as ei:
        gpsacq.Engine(6e6, 5e6)  # fc >= fs: the quadrature LO's quadrant index would leave its table (UB in the reference)
    assert ei.value.code == 1


def test_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import gpsacq
    with pytest.raises(gpsacq.GpsAcqError) as ei:
        gpsacq.Engine(4.092e6, 5.456e6, 5000.0)
    assert ei.value.code == 2 and "no CPU path" in str(ei.value)


def test_product_does_not_touch_oracle():
    """No file of the product (package, include/) mentions the oracle or the emulation harness."""
    bad = []
    for base in ("gnss-gps-sdr_amd", "include"):
        for dp, _, fns in os.walk(os.path.join(ROOT, base)):
            for fn in fns:
                if fn.endswith((".so", ".o", ".pyc")):
                    continue
                t = open(os.path.join(dp, fn), errors="replace").read()
                if re.search(r"oracle/|liboracle|oracle_lib|libemul", t):
                    bad.append(os.path.join(dp, fn))
    assert not bad, bad
    out = subprocess.run(["ldd", os.path.join(ROOT, "gnss-gps-sdr_amd", "lib", "libgpsacq.so")], capture_output=True, text=True).stdout
    assert "oracle" not in out and "emul" not in out and "amdhip64" in out


def test_handoff_matches_channel_start():
    """gpsacq_handoff against a direct restatement of c/channel.cpp:144-163 (host arithmetic)."""
    import math
    import numpy as np
    import gpsacq
    L1, CPS, N = 1575.42e6, 1.023e6, 40000
    for fc, fs, lo, ca, secs in [(4.092e6, 5.456e6, 6, 1465, 0.0), (4.092e6, 5.456e6, -9, 3868, 0.35),
                                 (2.6e6, 10e6, 12, 9999, 1.25), (0.62e6, 2.8e6, -700, 5, 0.01)]:
        pk = np.zeros(1, gpsacq.PEAK_DTYPE)[0]
        pk["lo_shift"], pk["ca_shift"], pk["snr"] = lo, ca, 100.0
        h = gpsacq.handoff(pk, fc, fs, secs)
        lo_dop = lo * fs / N
        ca_dop = lo_dop / L1 * CPS
        assert h["lo_dop_hz"] == lo_dop and h["ca_dop_hz"] == ca_dop
        assert h["lo_rate"] == int((fc + lo_dop) / fs * 2 ** 32) and h["ca_rate"] == int((CPS + ca_dop) / fs * 2 ** 32)
        spm = int(math.ceil(fs / 1000))
        ca2 = ca + int(np.rint(ca_dop * secs * fs / CPS))
        assert h["ca_shift"] == ca2 and h["ca_pause"] == (2 * spm - ca2) % spm
    # fs = 10 MHz reproduces the reference's literal (20000 - ca_shift) % 10000
    pk = np.zeros(1, gpsacq.PEAK_DTYPE)[0]
    pk["lo_shift"], pk["ca_shift"] = 3, 1234
    assert gpsacq.handoff(pk, 2.6e6, 10e6)["ca_pause"] == (20000 - 1234) % 10000


def _build_c_client(tmp_path):
    exe = str(tmp_path / "abi_smoke")
    libdir = os.path.join(ROOT, "gnss-gps-sdr_amd", "lib")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c", "abi_smoke.c"), "-o", exe, "-L", libdir, "-lgpsacq",
                           "-Wl,-rpath," + libdir])
    return exe


def test_c99_client_compiles_and_fails_loudly_without_gpu(tmp_path):
    """include/gpsacq.h is valid C99 (-pedantic -Werror) and a plain C program links against the library."""
    import torch
    exe = _build_c_client(tmp_path)
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu-marked test")
    r = subprocess.run([exe, os.path.join(ROOT, "tests", "golden", "gps_sig_tmp.bin"), "2.046e6", "8.184e6"], capture_output=True, text=True)
    assert r.returncode == 2 and "no CPU path" in r.stderr


@pytest.mark.gpu
def test_c99_client_on_gpu(tmp_path):
    exe = _build_c_client(tmp_path)
    r = subprocess.run([exe, os.path.join(ROOT, "tests", "golden", "gps_sig_tmp.bin"), "2.046e6", "8.184e6"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert r.stdout.startswith("bins 49 lags 8184 best sv 7 snr 713.6 lo_shift 0 ca_shift 260 doppler 0.0 Hz")
    assert "half-bin grid: 97 points of 102.30 Hz; multi (1 device) block 7 sv 7 snr 713.6 lo_shift 0 ca_shift 260" in r.stdout
    assert "pipeline: 16 + 16 peaks" in r.stdout and "multi blocks (2 engines on device 0): best sv 7 snr 713.6" in r.stdout
    assert "iq8: block 0 sv 7 snr" in r.stdout


def test_lds_dma_wait_is_in_the_isa(tmp_path):
    """k_corr<..., FOLD> refreshes its twiddle tables by LDS-DMA (`buffer_load ... lds`): every wave must drain its vmcnt before the
    barrier that lets the other waves read them.  The source spells the wait out; this compiles the kernels to gfx950 assembly
    (device side only, a few seconds) and checks that it is there, in front of the barrier, once per phase-1 copy."""
    hipcc = "/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else "hipcc"
    asm = str(tmp_path / "acq_kernels.s")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-function", "-Wno-unused-command-line-argument", "-S",
                           "--cuda-device-only", os.path.join(ROOT, "gnss-gps-sdr_amd", "csrc", "acq_kernels.hip"), "-o", asm])
    r = subprocess.run([os.sys.executable, os.path.join(ROOT, "tools", "isa_census.py"), asm, "--assert-dma-wait"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "LDS-DMA waits in front of their barriers: ok" in r.stdout
