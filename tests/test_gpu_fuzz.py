"""A bounded slice of tools/fuzz_gpu.py inside the GPU suite: seeded random configurations (sampling rate 1.2-20 MHz, IF, Doppler
range and step; coherent / reference quirk / non-coherent with and without creep re-alignment / windows / strides / 8-bit IQ in its
three sample modes) against the oracle's restatements, the pipeline and several engines on one GPU against a plain search bit for
bit, and the gps_test front end on random files byte for byte.  The wide runs are logged in profiles/r03f_fuzz.txt."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fuzz():
    spec = importlib.util.spec_from_file_location("fuzz_gpu", os.path.join(ROOT, "tools", "fuzz_gpu.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("seed", [8, 14, 1001, 1006, 1016, 2019] + list(range(100, 112)))
def test_random_configuration_against_the_oracle(seed):
    # 8, 14: creep re-alignment beyond 10000 lags (the gap the first wide run found); 1001/1006/1016: multi-bit, complex, sub-bin grids
    desc, worst = _fuzz().one(seed)
    assert worst <= 2e-5, desc


@pytest.mark.parametrize("seed", range(3000, 3008))
def test_random_plumbing_is_bit_exact(seed, monkeypatch):
    monkeypatch.setenv("FUZZ_PLUMBING", "1")
    desc, worst = _fuzz().one(seed)
    assert worst == 0.0, desc


@pytest.mark.parametrize("seed", range(4000, 4006))
def test_random_file_through_gps_test(seed, monkeypatch):
    monkeypatch.setenv("FUZZ_CLI", "1")
    desc, worst = _fuzz().one(seed)
    assert worst == 0.0, desc
