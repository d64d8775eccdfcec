"""GPU: round-4 additions -- the advisor's reserve/schedule item, every RCCL line executed on the one GPU (single-rank
communicator), the de-serialised multi-engine uploads, and the second reference-defined known answer (gps_sig_gen.m's
HackRF transmit file).  Identities between two routes through the HIP path are compared bit for bit."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def _nott(golden_dir, nblk=64):
    return open(os.path.join(golden_dir, "synth_nott_fs5456.bin"), "rb").read()[:nblk * 5120]


@pytest.mark.parametrize("quirks", [False, True])
def test_reserve_between_searches_keeps_the_schedule_valid(golden_dir, quirks):
    """search -> gpsacq_reserve(larger) -> search (round-3 advisor finding): the reserve re-allocates the task list (and, with
    ref_quirks, the patched code slots and patch list); the cached reference schedule must be rebuilt, not reused from freed
    memory.  Both searches, and a shorter one afterwards (a prefix of the cached schedule), are bit-exact."""
    import gpsacq
    buf = np.frombuffer(_nott(golden_dir, 96), dtype=np.uint8)
    with gpsacq.Engine(4.092e6, 5.456e6, 5000.0, ref_quirks=quirks) as ref:
        _, want = ref.search(buf, want_cells=False)
    with gpsacq.Engine(4.092e6, 5.456e6, 5000.0, ref_quirks=quirks) as eng:
        _, a = eng.search(buf[:33 * 5120], want_cells=False)  # caches a 33-task schedule
        assert np.array_equal(a, want[:33])
        eng.reserve(4096)                                      # grows every buffer the schedule lives in
        _, b = eng.search(buf[:32 * 5120], want_cells=False)   # n_tasks <= the old sched_tasks: the old fast path
        assert np.array_equal(b, want[:32])
        _, c = eng.search(buf, want_cells=False)
        assert np.array_equal(c, want)
        eng.reserve(8192)
        eng.reserve(64)                                        # smaller: nothing moves
        _, d = eng.search(buf[:65 * 5120], want_cells=False)
        assert np.array_equal(d, want[:65])


def test_iq8_quirk_route_checks_the_stride_before_it_converts():
    """8-bit IQ + ref_quirks converts 40960 samples per block before it searches (round-3 advisor finding): a stride that holds
    fewer is refused up front -- before the conversion kernel is enqueued on a buffer that short -- not after."""
    import gpsacq
    iq = np.full(2 * 80000, 128, dtype=np.uint8)
    with gpsacq.Engine(0.62e6, 2.8e6, 5000.0, ref_quirks=True) as eng:
        with pytest.raises(gpsacq.GpsAcqError, match="81920"):
            eng.search_iq8(iq, eng.iq8_input(signed=False, remove_dc=False), stride=80000)
        iq2 = np.full(2 * 81920, 128, dtype=np.uint8)
        iq2[::7] = 140
        _, pk = eng.search_iq8(iq2, eng.iq8_input(signed=False, remove_dc=False), stride=81920)  # the full stride still works
        assert pk.shape == (2,)


# ---- every RCCL line on the one GPU --------------------------------------------------------------------------------------
def _key(p):
    return (float(p["snr"]), -int(p["lo_shift"]), int(p["ca_shift"]))


@pytest.mark.parametrize("devices", [(0,), (0, 0, 0)])
def test_rccl_single_rank_multi_search(golden_dir, devices, monkeypatch):
    """GPSACQ_MULTI_FORCE_RCCL=1: gpsacq_multi_create goes through dlopen(librccl) + ncclCommInitAll over the one distinct GPU and
    both merges issue the real ncclAllReduce(ncclUint64 / ncclFloat32, ncclMax) inside ncclGroupStart/End -- the calls of an
    8-GPU node, as a one-rank group.  Results: bit-equal to the plain search, max_pwr included, in both decompositions; also
    with three engines sharing the GPU (device-side merge first, then the one-rank all-reduce of the representative)."""
    import gpsacq
    buf = _nott(golden_dir)
    with gpsacq.Engine(4.092e6, 5.456e6, 5000.0) as eng:
        _, want = eng.search(buf, want_cells=False)
        tasks = [(b, sv) for b in (0, 33) for sv in range(32)]
        _, wgrid = eng.search(buf, tasks=tasks, want_cells=False)
    monkeypatch.setenv("GPSACQ_MULTI_FORCE_RCCL", "1")
    with gpsacq.MultiEngine(4.092e6, 5.456e6, 5000.0, devices=devices) as me:
        assert me.last_call_ms()["rccl_allreduces"] == 0
        for rep in range(2):
            peaks, best = me.search_blocks(buf)
            assert np.array_equal(peaks, want)
            for sv in range(32):
                cand = want[sv::32]
                k = max(_key(p) for p in cand)
                assert _key(best[sv]) == k
                w = [p for p in cand if _key(p) == k][0]
                assert float(best["max_pwr"][sv]) == float(w["max_pwr"])
            t = me.last_call_ms()
            assert t["rccl_allreduces"] == 4 * rep + 2  # keys + winner powers
            assert 0 < t["enqueue_ms"] <= t["total_ms"]
            got = me.search_grid(buf, tasks)
            assert np.array_equal(got, wgrid)
            assert me.last_call_ms()["rccl_allreduces"] == 4 * rep + 4
    monkeypatch.delenv("GPSACQ_MULTI_FORCE_RCCL")
    with gpsacq.MultiEngine(4.092e6, 5.456e6, 5000.0, devices=devices) as me:  # without the switch: no communicator, no RCCL call
        peaks, _ = me.search_blocks(buf)
        assert np.array_equal(peaks, want) and me.last_call_ms()["rccl_allreduces"] == 0


def _bench(*args):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GPSACQ_DIST_BACKEND"):
        e.pop(k, None)
    if "--live-traffic" not in args:
        args = ("--no-live-traffic",) + tuple(args)  # its own flag since round 5 (it used to follow --no-cpu-baseline)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-e2e", *args],
                       capture_output=True, text=True, timeout=600, env=e)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_rccl_single_rank_bench_line():
    """bench.py at N = 1 joins a one-rank nccl (= RCCL) process group: Leg.step's nccl all-reduce of the packed keys, the
    barriers of the fences and the device-side all-reduce of the elapsed time all run; the merged keys still name the
    injected satellites.  --force-dist: no silent fallback.  The soak leg runs after the timed steps and leaves them alone."""
    j = _bench("--force-dist", "--steps", "3", "--warmup", "1", "--blocks-total", "640", "--weak-blocks", "0", "--soak-seconds", "1.5")
    assert j["dist_backend"] == "nccl" and j["rccl_ranks_seen"] == 1 and j["n_gpus"] == 1 and j["dist_note"] is None
    assert j["steps"] == 3 and j["blocks_per_rank"] == [640]
    assert set(j["detected_prns"]) >= set(j["injected_prns_all_ranks"])
    sk = j["soak"]
    assert sk["steps"] >= 20 and sk["seconds"] >= 1.5 and sk["ms_per_step"] > 0
    assert abs(sk["ms_per_step"] / j["ms_per_step"] - 1) < 0.5  # the same step (a 3-step timed region is noisy; not a perf assertion)
    sh = j["strong_share_at_8"]  # one GPU's share of the capture at N = 8: 3 of the 20 runs
    assert sh["blocks_per_step"] == 96 and sh["ms_per_step"] > 0 and sh["predicted_speedup_at_8_gpus"] > 1
    j0 = _bench("--no-dist", "--steps", "2", "--warmup", "1", "--blocks-total", "640", "--weak-blocks", "0", "--soak-seconds", "0")
    assert j0["dist_backend"] is None and "soak" not in j0 and j0["detected_prns"] == j["detected_prns"]


def test_live_traffic_is_measured_in_the_run():
    """roofline.traffic of the default line comes from two rocprofv3 --pmc child runs of the same command on the same box (not
    from the committed profile): a few KB per cell -- both spectra of a cell come out of L2 -- far below the 1.28 MB of
    algorithmic bytes."""
    j = _bench("--live-traffic", "--steps", "2", "--warmup", "1", "--blocks-total", "640", "--weak-blocks", "0", "--soak-seconds", "0")
    r = j["roofline"]
    assert r["traffic_live"] is not None and "error" not in r["traffic_live"], r["traffic_live"]
    assert r["traffic_stale"] is False and r["traffic_source"].startswith("live (estimated correction)")
    per_cell = r["traffic"] / r["cells_per_launch"]
    assert 100 < per_cell < 100000, per_cell


def test_multi_enqueue_time_is_flat_in_the_number_of_engines(golden_dir):
    """The calling thread is out of the devices' critical path: with 1, 2, 4, 8 engines (sharing the one GPU here) the host
    time until everything is enqueued does not grow with the engine count -- each engine's staging copy and enqueue run on
    its own thread.  (Round 3: pageable hipMemcpyAsync on one thread, every device behind the copies of the ones before it.)"""
    import gpsacq
    one = np.frombuffer(_nott(golden_dir, 64), dtype=np.uint8)
    buf = np.tile(one, 16)  # 32 runs, 5.2 MB
    rows = {}
    with gpsacq.Engine(4.092e6, 5.456e6, 5000.0) as eng:
        _, want = eng.search(buf, want_cells=False)
    for n in (1, 2, 4, 8):
        with gpsacq.MultiEngine(4.092e6, 5.456e6, 5000.0, devices=(0,) * n) as me:
            me.search_blocks(buf)  # buffers, schedules, kernels
            ts = []
            for _ in range(3):
                peaks, _ = me.search_blocks(buf)
                ts.append(me.last_call_ms())
            assert np.array_equal(peaks, want)
            rows[n] = min(t["enqueue_ms"] for t in ts), min(t["total_ms"] for t in ts)
    print("engines: (enqueue ms, total ms)", rows)
    # generous: thread start-up and 8 x the launches on one GPU's queues; the serial form grew by the whole copy + search per engine
    assert rows[8][0] < rows[1][0] + 0.5 * rows[1][1], rows


# ---- gps_sig_gen.m:21-30: the HackRF transmit file, generated on the device and searched as complex baseband -------------
def _chips(prn):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_lib import lib, _p
    c = np.zeros(1023, np.uint8)
    lib("f64").oracle_ca_chips(prn - 1, _p(c))
    return c


def test_device_tx_generator_equals_the_restatement_and_is_found_as_complex_baseband(golden_dir, tmp_path):
    """gpsacq_generate_sig_tx (k_siggen_tx) against oracle/sig_gen_oracle.py::hackrf_tx, bit for bit, on the head of the stream,
    the seam between two repetitions and the filter tail; then the script's known answer through the product's complex-baseband
    path (int8, no DC removal, IF 0, fs 8.184 MHz): PRN 8 in every block, Doppler bin 0 +- 1, ca_shift = (40960 b - 20) mod 8184
    -- the law of the 1-bit file (384 blocks pinned there), here in the second file format; and the front end on the file."""
    import gpsacq
    import sig_gen_oracle as sg
    db = json.load(open(os.path.join(golden_dir, "gps_sig_tmp_databits.json")))
    prn, data = db["prn"], db["bits_pm1"]
    chips = _chips(prn)
    total, per_rep = sg.tx_samples(len(data)), 16368000
    with gpsacq.Engine(0.0, 8.184e6, 5000.0) as eng:
        for first, count in ((0, 2 * 1310720), (per_rep - 300000, 600000), (total - 100000, 100000)):
            got = eng.generate_sig_tx(prn, data, first_sample=first, n_samples=count)
            want = sg.hackrf_tx(chips, data, first=first, count=count)
            assert got.dtype == np.int8 and np.array_equal(got, want), (first, int((got != want).sum()))
        with pytest.raises(gpsacq.GpsAcqError):
            eng.generate_sig_tx(prn, data, first_sample=total - 10, n_samples=11)
        # two runs of the file (64 blocks of 40960 complex samples) + the blocks around the repetition seam
        nblk = 64
        iq = eng.generate_sig_tx(prn, data, first_sample=0, n_samples=nblk * 40960)
        inp = eng.iq8_input(signed=True, remove_dc=False, total_samples=iq.size // 2, multibit=2)
        tasks = [(b, prn - 1) for b in range(nblk)] + [(0, sv) for sv in range(32)]
        _, pk = eng.search_iq8(iq, inp, tasks=tasks)
        for b in range(nblk):
            assert pk["snr"][b] > 300 and abs(int(pk["lo_shift"][b])) <= 1 and int(pk["ca_shift"][b]) == (40960 * b - 20) % 8184, (b, pk[b])
        blk0 = pk[nblk:]
        assert int(np.argmax(blk0["snr"])) == prn - 1 and np.sort(blk0["snr"])[-2] < 40
        seam = eng.generate_sig_tx(prn, data, first_sample=398 * 40960, n_samples=4 * 40960)
        _, ps = eng.search_iq8(seam, eng.iq8_input(signed=True, remove_dc=False, total_samples=seam.size // 2, multibit=2), tasks=[(i, prn - 1) for i in range(4)])
        for i in range(4):  # (block 399 holds the bit flip of the seam: its broad peak sits one lag early, in the float64 restatement too)
            assert ps["snr"][i] > 300 and int(ps["ca_shift"][i]) == (40960 * (398 + i) - 20) % 8184 - (1 if i == 1 else 0)
        # reference schedule on the file: gps_test's report names sv 7 in both runs, at the ca_shift of block 32 r + 7
        path = str(tmp_path / "gps_sig_tmp_for_hackrf_tx.head.bin")
        iq.tofile(path)
    exe = os.path.join(ROOT, "gnss-gps-sdr_amd", "bin", "gps_test")
    r = subprocess.run([exe, path, "0", "8.184e6", "5000"], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, GPSACQ_INPUT="iq_s8", GPSACQ_IQ_COMPLEX="1", GPSACQ_IQ_KEEP_DC="1"))
    assert r.returncode == 0, r.stderr
    lines = r.stdout.splitlines()
    for run in range(2):
        sat = [l for l in lines if l.startswith("%2d satellite:" % run)][0].split(":")[1].split()
        ca = [l for l in lines if l.startswith("%2d  ca_shift:" % run)][0].split(":")[1].split()
        lo = [l for l in lines if l.startswith("%2d  lo_shift:" % run)][0].split(":")[1].split()
        k = sat.index("7")
        assert int(ca[k]) == (40960 * (32 * run + 7) - 20) % 8184 and abs(int(lo[k])) <= 1
    assert "run out of file!" in r.stdout
