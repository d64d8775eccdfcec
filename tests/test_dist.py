"""N > 1 path on CPU: world_size-2 gloo processes exercise the sharding helpers and the one
collective of the path (all-reduce(MAX) of the packed per-PRN peaks).  The reference has no
multi-process code; the expected values are the single-process results."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_peaks(n, seed, dmax=36):
    rng = np.random.default_rng(seed)
    pk = np.zeros(n, dtype=[("snr", "<f4"), ("lo", "<i4"), ("ca", "<i4"), ("mp", "<f4")])
    pk["snr"] = (rng.random(n) * 120).astype(np.float32)
    pk["snr"][::7] = np.float32(50.0)  # force exact SNR ties across ranks
    pk["lo"] = rng.integers(-dmax, dmax + 1, n)
    pk["ca"] = rng.integers(0, 5456, n)
    pk["mp"] = pk["snr"] * 1e14
    return pk


def _worker(rank, world, port, n_runs, q):
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gnss-gps-sdr_amd", "python"))
    from gpsacq import dist as D
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # --- decomposition by block: contiguous runs per rank ---
        allpk = _fake_peaks(n_runs * 32, 123)
        first, n = D.shard_runs(n_runs, rank, world)
        mine = torch.from_numpy(allpk[first * 32:(first + n) * 32].copy().view(np.int32).reshape(-1, 4))
        best = D.per_prn_best(D.pack_keys(mine, 36)) if n > 0 else torch.zeros(32, dtype=torch.int64)
        best = D.allreduce_best(best)
        gathered = D.gather_peaks(mine)
        # --- decomposition by Doppler slab: per-cell SNRs of one (block, PRN) from the golden file ---
        z = np.load(os.path.join(GOLDEN, "np64_cells_nott.npz"))
        dmax, S = int(z["dmax"]), int(z["S"])
        slab_best = torch.zeros(32, dtype=torch.int64)
        f, nb = D.shard_doppler(dmax, rank, world)
        for b, sv in [tuple(int(v) for v in p) for p in z["pairs"]][:5]:
            snr = (z[f"max_pwr_{b}_{sv}"] / (z[f"tot_pwr_{b}_{sv}"] / S)).astype(np.float32)
            mi = z[f"max_i_{b}_{sv}"]
            p = np.zeros(1, dtype=[("snr", "<f4"), ("lo", "<i4"), ("ca", "<i4"), ("mp", "<f4")])
            for d in range(f, f + nb):  # Correlate's scan over this rank's slab, strict '>'
                if snr[d + dmax] > p["snr"][0]:
                    p["snr"][0], p["lo"][0], p["ca"][0] = snr[d + dmax], d, mi[d + dmax]
            slab_best[sv] = D.pack_keys(torch.from_numpy(p.view(np.int32).reshape(1, 4)), dmax)[0]
        slab_best = D.allreduce_best(slab_best)
        if rank == 0:
            q.put((best.numpy(), gathered.numpy(), slab_best.numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_runs", [5, 1])
def test_two_rank_gloo(n_runs):
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gnss-gps-sdr_amd", "python"))
    from gpsacq import dist as D
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_runs, q)) for r in range(2)]
    for p in procs:
        p.start()
    best, gathered, slab_best = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # expected: single-process results
    allpk = _fake_peaks(n_runs * 32, 123)
    t = torch.from_numpy(allpk.copy().view(np.int32).reshape(-1, 4))
    exp = D.per_prn_best(D.pack_keys(t, 36))
    assert np.array_equal(best, exp.numpy())
    assert np.array_equal(gathered, t.numpy())
    snr, lo, ca = D.unpack_keys(torch.from_numpy(best), 36)
    for sv in range(32):  # reference order: highest SNR, ties to the lowest Doppler bin
        cand = allpk[sv::32]
        top = cand[cand["snr"] == cand["snr"].max()]
        assert snr[sv].item() == top["snr"][0] and lo[sv].item() == top["lo"].min()
    # Doppler slabs: equal to the full-range sequential scan
    z = np.load(os.path.join(GOLDEN, "np64_cells_nott.npz"))
    dmax, S = int(z["dmax"]), int(z["S"])
    s2, lo2, ca2 = D.unpack_keys(torch.from_numpy(slab_best), dmax)
    for b, sv in [tuple(int(v) for v in p) for p in z["pairs"]][:5]:
        snr_c = (z[f"max_pwr_{b}_{sv}"] / (z[f"tot_pwr_{b}_{sv}"] / S)).astype(np.float32)
        d = int(np.argmax(snr_c))  # first maximum == strict '>' scan
        assert lo2[sv].item() == d - dmax and ca2[sv].item() == int(z[f"max_i_{b}_{sv}"][d]) and s2[sv].item() == snr_c[d]


def test_shard_helpers():
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gnss-gps-sdr_amd", "python"))
    from gpsacq import dist as D
    for n in (0, 1, 7, 64, 341):
        for w in (1, 2, 4, 8):
            parts = [D.shard_runs(n, r, w) for r in range(w)]
            assert sum(p[1] for p in parts) == n
            assert all(parts[i][0] + parts[i][1] == parts[i + 1][0] for i in range(w - 1))
            assert max(p[1] for p in parts) - min(p[1] for p in parts) <= 1
    assert D.shard_doppler(36, 0, 2) == (-36, 37) and D.shard_doppler(36, 1, 2) == (1, 36)
    # Doppler grids of any step (gpsacq_set_doppler_step): configs[4]'s 4399 points over 8 ranks, and more ranks than points
    parts = [D.shard_doppler_grid(4399, -2199, r, 8) for r in range(8)]
    assert parts[0][0] == -2199 and parts[-1][0] + parts[-1][1] - 1 == 2199 and sum(p[1] for p in parts) == 4399
    assert all(parts[i][0] + parts[i][1] == parts[i + 1][0] for i in range(7))
    parts = [D.shard_doppler_grid(3, -1, r, 8) for r in range(8)]
    assert [p[1] for p in parts] == [1, 1, 1, 0, 0, 0, 0, 0]
    # strong scaling of bench.py: 340 runs (the Nottingham capture) over 8 ranks
    assert [D.shard_runs(340, r, 8)[1] for r in range(8)] == [43, 43, 43, 43, 42, 42, 42, 42]


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench_cli(*args, env=None):
    import subprocess
    import sys
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=300, env=e)


def test_bench_gpus_n_spawns_its_own_ranks():
    """`python bench.py --gpus 2` without WORLD_SIZE must not measure one GPU and call it two: it re-launches
    itself under torch.distributed.run; --spawn-check stops every rank after the rendezvous so this runs without a GPU."""
    import json
    r = _bench_cli("--gpus", "2", "--spawn-check", env={"GPSACQ_DIST_BACKEND": "gloo"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["rccl_ranks_seen"] == 2 and j["world_size_env"] == 2
    # one rank stays one process
    r = _bench_cli("--spawn-check")
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert j["n_gpus"] == 1 and j["rccl_ranks_seen"] == 1


def test_bench_gpus_n_without_the_devices_fails_loudly():
    """RCCL backend, fewer visible devices than --gpus: non-zero exit and no JSON line (here: no GPU at all)."""
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 8:
        pytest.skip("8 devices visible")
    r = _bench_cli("--gpus", "8")
    assert r.returncode != 0
    assert "--gpus 8" in r.stderr and "visible" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
    # a WORLD_SIZE that contradicts --gpus is refused too (both directions)
    r = _bench_cli("--gpus", "1", "--spawn-check", env={"WORLD_SIZE": "2", "RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr



def test_merged_keys_do_not_depend_on_the_world_size():
    """What bench.py's `keys_digest` rests on (north_star: "identical acquisition results ... at 1/2/4/8"): the per-PRN best keys of the
    whole capture are the same 32 numbers however the 340 runs are dealt to 1, 2, 3, 4, 8 or 400 ranks (shard_runs: contiguous, balanced,
    ranks without work contribute zeros) -- integer MAX is associative, and ties (equal SNR) go to the lower Doppler bin whatever the
    order of the merges (c/search_offline.cpp:196-198)."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "gnss-gps-sdr_amd", "python"))
    sys.path.insert(0, root)
    from gpsacq import dist as D
    import bench_extras as X
    n_runs = 340
    allpk = _fake_peaks(n_runs * 32, 321)
    digests = {}
    for world in (1, 2, 3, 4, 8, 400):
        merged = torch.zeros(32, dtype=torch.int64)
        covered = 0
        for rank in range(world):
            first, n = D.shard_runs(n_runs, rank, world)
            assert first == covered  # contiguous, in rank order
            covered += n
            mine = torch.from_numpy(allpk[first * 32:(first + n) * 32].copy().view(np.int32).reshape(-1, 4))
            best = D.per_prn_best(D.pack_keys(mine, 36)) if n > 0 else torch.zeros(32, dtype=torch.int64)
            merged = torch.maximum(merged, best)
        assert covered == n_runs
        digests[world] = X.keys_digest(merged.numpy())
    assert len(set(digests.values())) == 1, digests
    # the digest sees a single changed code phase
    other = allpk.copy()
    win = int(np.argmax(other["snr"]))
    other["ca"][win] = (other["ca"][win] + 1) % 5456
    k = D.per_prn_best(D.pack_keys(torch.from_numpy(other.view(np.int32).reshape(-1, 4)), 36))
    assert X.keys_digest(k.numpy()) != digests[1]
