"""CPU check of the HIP kernels' index math: tests/emul runs the per-thread phase functions of
gnss-gps-sdr_amd/csrc/acq_phases.hpp on the host (test infrastructure) and is compared with
the oracle.  The real kernels are checked on the GPU by tests/test_gpu_parity.py."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from oracle_lib import Oracle, lib, _p

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def emul():
    path = os.path.join(ROOT, "tests", "emul", "libemul_acq.so")
    if not os.path.exists(path):
        subprocess.check_call(["make", "-C", ROOT, "emul"], stdout=subprocess.DEVNULL)
    E = ctypes.CDLL(path)
    vp, d, i = ctypes.c_void_p, ctypes.c_double, ctypes.c_int
    E.emul_forward_bits.argtypes = [vp] * 4
    E.emul_forward_bits_sub.argtypes = [vp, vp, vp, i, i, vp]
    E.emul_forward_real.argtypes = [vp] * 2
    E.emul_cell.argtypes = [vp, vp, i, i, i, i, i, i, vp, vp, vp]
    E.emul_code_replica.argtypes = [d, i, vp]
    E.emul_lo_masks.argtypes = [d, d, vp, vp]
    E.emul_search_code.argtypes = [i, i]
    E.emul_ca_chips.argtypes = [i, vp]
    E.emul_handout_walk.argtypes = [i, i, i, i, ctypes.c_uint, i, vp, vp]
    E.emul_handout_plan.argtypes = [i, vp, vp]
    E.emul_dmax.argtypes = [d, d]
    E.emul_nlags.argtypes = [d]
    return E


def test_host_tables_bit_exact(emul):
    L = lib("f64")
    for fs in (5.456e6, 8.184e6, 2.8e6, 10e6):
        for sv in (0, 7, 31):
            a, b = np.zeros(40000, np.float32), np.zeros(40000, np.float32)
            L.oracle_code_replica(fs, sv, _p(a))
            emul.emul_code_replica(fs, sv, _p(b))
            assert np.array_equal(a, b)
    for fc, fs in ((4.092e6, 5.456e6), (2.046e6, 8.184e6), (0.62e6, 2.8e6)):
        cosm, sinm = np.zeros(5120, np.uint8), np.zeros(5120, np.uint8)
        emul.emul_lo_masks(fc, fs, _p(cosm), _p(sinm))
        quad = np.zeros(40960, np.uint8)
        L.oracle_lo_quadrants(fc, fs, 40960, _p(quad))
        assert np.array_equal(np.unpackbits(cosm, bitorder="little"), np.array([0, 1, 1, 0], np.uint8)[quad])
        assert np.array_equal(np.unpackbits(sinm, bitorder="little"), np.array([1, 1, 0, 0], np.uint8)[quad])
        assert emul.emul_dmax(fs, 5000.0) == L.oracle_dmax(fs, 5000.0)
        assert emul.emul_nlags(fs) == L.oracle_nlags(fs)


def test_product_ca_generator_against_is_gps_200_table_3_I(emul, golden_dir):
    """The PRODUCT's C/A generator (acq_tables.hpp: CaCode, an integer shift-register pair, + kTaps) for all 32 PRNs against
    the reference-held IS-GPS-200G Table 3-I: first ten chips (octal column) and the whole period = G1 xor G2 delayed by the
    table's chip delay (c/cacode.h:9-35, c/search_offline.cpp:20-53), and against the oracle chip for chip."""
    from test_oracle import _table_3_I, _g1_g2
    first10, delays = _table_3_I(golden_dir)
    g1, g2 = _g1_g2()
    L = lib("f64")
    for sv in range(32):
        chips, o = np.zeros(1023, np.uint8), np.zeros(1023, np.uint8)
        emul.emul_ca_chips(sv, _p(chips))
        L.oracle_ca_chips(sv, _p(o))
        assert list(chips[:10]) == first10[sv], f"PRN {sv + 1}"
        assert np.array_equal(chips, g1 ^ np.roll(g2, delays[sv])), f"PRN {sv + 1}"
        assert np.array_equal(chips, o)


@pytest.mark.parametrize("fc,fs,file,mc,w1h", [(4.092e6, 5.456e6, "synth_nott_fs5456.bin", 22, 0), (2.046e6, 8.184e6, "gps_sig_tmp.bin", 33, 1),
                                               (0.62e6, 2.8e6, "synth_rtl_fs2800.bin", 12, 0), (4.092e6, 5.456e6, "synth_nott_fs5456.bin", 28, 1)])
def test_emulated_kernels_vs_oracle(emul, golden_dir, fc, fs, file, mc, w1h):
    buf = open(os.path.join(golden_dir, file), "rb").read()
    blk = np.frombuffer(buf[7 * 5120:8 * 5120], np.uint8).copy()
    orc = Oracle(fc, fs, 5000.0)
    cosm, sinm = np.zeros(5120, np.uint8), np.zeros(5120, np.uint8)
    emul.emul_lo_masks(fc, fs, _p(cosm), _p(sinm))
    # Sample()
    d_emul = np.zeros(80000, np.float32)
    emul.emul_forward_bits(_p(blk), _p(cosm), _p(sinm), _p(d_emul))
    d_orc = orc.sample_spectrum(blk)
    assert np.abs(d_emul.view(np.complex64) - d_orc).max() / np.abs(d_orc).max() < 2e-6
    # SearchInit()
    rep = np.zeros(40000, np.float32)
    emul.emul_code_replica(fs, 7, _p(rep))
    c_emul = np.zeros(80000, np.float32)
    emul.emul_forward_real(_p(rep), _p(c_emul))
    c_orc = orc.code_spectrum(7)
    assert np.abs(c_emul.view(np.complex64) - c_orc).max() / np.abs(c_orc).max() < 2e-6
    # Correlate() cells at the edges and the middle of the Doppler range
    cells, _ = orc.search_block(blk, 7)
    d_in = np.ascontiguousarray(d_orc).view(np.float32)
    c_in = np.ascontiguousarray(c_orc).view(np.float32)
    assert emul.emul_lane_maps_are_permutations() == 0
    for dop in (-orc.dmax, -9, 0, 1, orc.dmax):
        got = {}
        for lay in (1, 2, 3):  # LayB (round 2's lane map), LayC (conflict-free lanes), LayC + folded rotation
            mp, mi, tp = ctypes.c_float(), ctypes.c_int(), ctypes.c_float()
            assert emul.emul_cell(_p(d_in), _p(c_in), 24, dop, orc.num_lags, mc, w1h, lay, ctypes.byref(mp), ctypes.byref(mi), ctypes.byref(tp)) == 0
            ref = cells[dop + orc.dmax]
            assert abs(mp.value / ref["max_pwr"] - 1) < 2e-5 and abs(tp.value / ref["tot_pwr"] - 1) < 2e-5
            assert mi.value == ref["max_i"]
            got[lay] = (mp.value, mi.value, tp.value)
        # the lane maps only re-deal the same butterflies: the peak is bit-identical, the power sum differs by its order at most
        assert got[1][:2] == got[2][:2] and abs(got[1][2] / got[2][2] - 1) < 1e-6
        # the folded rotation multiplies by one table value where the other form multiplies by two: powers agree to float rounding
        assert got[3][1] == got[2][1] and abs(got[3][0] / got[2][0] - 1) < 2e-6 and abs(got[3][2] / got[2][2] - 1) < 2e-6

def test_emulated_subbin_forward_vs_oracle(emul, golden_dir):
    """Sub-bin Doppler offsets folded into the forward transform's twiddles (gpsacq_set_doppler_step) against the
    oracle's restatement: Sample() of the block multiplied by exp(-2 pi i (r/R) n / N)."""
    fc, fs = 4.092e6, 5.456e6
    buf = open(os.path.join(golden_dir, "synth_nott_fs5456.bin"), "rb").read()
    blk = np.frombuffer(buf[3 * 5120:4 * 5120], np.uint8).copy()
    orc = Oracle(fc, fs, 5000.0)
    cosm, sinm = np.zeros(5120, np.uint8), np.zeros(5120, np.uint8)
    emul.emul_lo_masks(fc, fs, _p(cosm), _p(sinm))
    for sub, r in ((3, 1), (3, 2), (4, 3)):
        d_emul = np.zeros(80000, np.float32)
        emul.emul_forward_bits_sub(_p(blk), _p(cosm), _p(sinm), sub, r, _p(d_emul))
        orc.L.oracle_sample_ramped(orc.h, _p(blk), float(r) / sub)
        d_orc = np.zeros(80000, np.float32)
        orc.L.oracle_get_sample_spectrum(orc.h, _p(d_orc))
        a, b = d_emul.view(np.complex64), d_orc.view(np.complex64)
        assert np.abs(a - b).max() / np.abs(b).max() < 3e-6, (sub, r)
    # r = 0 is the reference's Sample()
    d0, d1 = np.zeros(80000, np.float32), np.zeros(80000, np.float32)
    emul.emul_forward_bits_sub(_p(blk), _p(cosm), _p(sinm), 3, 0, _p(d0))
    emul.emul_forward_bits(_p(blk), _p(cosm), _p(sinm), _p(d1))
    assert np.array_equal(d0, d1)
    d_orc = orc.sample_spectrum(blk)
    assert np.abs(d1.view(np.complex64) - d_orc).max() / np.abs(d_orc).max() < 2e-6

def test_lane_maps_are_bank_conflict_free_in_the_lds_model(emul):
    """The lane maps the kernels really use (read out of acq_math.hpp through the emulation library) in the instruction-level
    LDS model of tools/lds_maps.py: round 2's map (LayB) shows the 488 conflict cycles per sub-transform the hardware counter
    measured (SQ_LDS_BANK_CONFLICT, 3904 per cell), the product's (LayC) at most 20, and a quarter fewer LDS cycles."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import lds_maps
    out = {}
    for lay in (1, 2):
        r = lds_maps.model(lambda t: emul.emul_pass1_jp(lay, t),
                           lambda e: divmod(emul.emul_pass2_owner(lay, e), 100),
                           lambda t: emul.emul_pass3_rho(lay, t))
        out[lay] = (sum(v[0] for v in r.values()), sum(v[1] for v in r.values()))
    assert out[1] == (1930, 488)
    assert out[2][1] <= 20 and out[2][0] <= 1462
    assert [emul.emul_pass3_rho(2, t) for t in range(250)] == lds_maps.make_rho_c()  # the committed table is the generator's



def test_handout_plan_of_the_baseline_grids(emul):
    """acq_phases.hpp handout_plan: the reference's grids (73 / 49 / 143 bins) are one unit per task; fine grids are cut into chunks of
    <= 128 points that cover the task with fewer void tickets than units."""
    u, c = ctypes.c_int(), ctypes.c_int()
    for ndop, units, chunk in ((1, 1, 1), (49, 1, 49), (73, 1, 73), (143, 1, 143), (146, 1, 146), (147, 2, 74), (219, 2, 110), (953, 8, 120), (2857, 23, 125), (4399, 35, 126)):
        emul.emul_handout_plan(ndop, ctypes.byref(u), ctypes.byref(c))
        assert (u.value, c.value) == (units, chunk), ndop
        assert c.value <= 146 and units * chunk >= ndop and units * chunk - ndop < units


@pytest.mark.parametrize("n_tasks,ndop,wgs,n_xcd,skew", [(340, 73, 768, 8, 0), (64, 73, 768, 8, 3), (1, 73, 768, 8, 0), (5, 1, 96, 8, 1), (12, 2857, 768, 8, 2),
                                                        (128, 953, 768, 8, 0), (7, 4399, 768, 8, 5), (3, 147, 24, 3, 0), (40, 49, 768, 1, 0), (1000, 3, 50, 8, 4)])
def test_handout_gives_every_cell_to_exactly_one_workgroup(emul, n_tasks, ndop, wgs, n_xcd, skew):
    """The hand-out of k_corr<..., PERSIST> walked on the CPU with the product's index arithmetic, workgroups scheduled in random order
    (several seeds; XCDs of unequal speed): every (task, Doppler point) is handed out exactly once, every workgroup leaves, the slot
    table's bound holds, a faster XCD ends up with more units, and waits on an unpublished slot do occur (the kernel's spin is needed)."""
    total_waits = 0
    for seed in range(4):
        counts = np.zeros(n_tasks * ndop, np.uint32)
        stats = np.zeros(8, np.int64)
        rc = emul.emul_handout_walk(n_tasks, ndop, wgs, n_xcd, seed, skew, counts.ctypes.data_as(ctypes.c_void_p), stats.ctypes.data_as(ctypes.c_void_p))
        assert rc == 0, (rc, seed)
        assert (counts == 1).all(), (seed, int((counts != 1).sum()))
        assert stats[0] == n_tasks * ndop and stats[3] <= stats[4]
        u, c = ctypes.c_int(), ctypes.c_int()
        emul.emul_handout_plan(ndop, ctypes.byref(u), ctypes.byref(c))
        assert stats[1] >= 0 and (u.value * c.value == ndop or stats[1] > 0 or n_tasks == 0)
        if skew >= 3 and n_xcd == 8 and n_tasks * u.value >= 64:
            assert stats[6] > stats[5], (int(stats[5]), int(stats[6]))  # the XCD scheduled most often took more units than the slowest
        total_waits += int(stats[2])
    if n_tasks * ndop > 4 * wgs and n_xcd > 1:
        assert total_waits > 0
