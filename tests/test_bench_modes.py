"""bench.py's other modes on the GPU box (the driver only runs the default line): a capture file instead of synthetic data,
and the configs[4] grid with a Doppler step.  Short runs; the numbers are not asserted, the results are."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*args, env=None):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    if "--soak-seconds" not in args:
        args = ("--soak-seconds", "0") + tuple(args)  # the soak leg has its own test (tests/test_gpu_round4.py)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-live-traffic", *args],
                       capture_output=True, text=True, timeout=600, env=e)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_capture_mode_on_the_bundled_file(golden_dir):
    """`bench.py --capture` (the hook for gps.samples.1bit.I.fs5456.if4092.bin) on the reference's gps_sig_tmp.bin:
    12 whole runs, PRN 8 at zero Doppler (README.md:45,57)."""
    j = _bench("--config", "2", "--capture", os.path.join(golden_dir, "gps_sig_tmp.bin"))
    assert j["data"] == "capture file" and j["config"]["cells_per_step_job"] == 12 * 32 * 49
    assert j["scaling"] == "strong" and j["roofline"]["bound"] == "valu_fp32" and 0 < j["roofline"]["frac"] < 1
    best = {d["prn"]: d for d in j["detected"]}
    assert 8 in best and best[8]["lo_shift"] == 0 and best[8]["snr"] > 500


def test_configs4_line():
    j = _bench("--config", "4", "--doppler-step", "50", "--grid-blocks", "1")
    assert "4399 Doppler points" in j["config"]["workload"] and j["config"]["cells_per_step_job"] == 32 * 4399
    assert j["value"] > 0 and j["roofline"]["kernel"] == "k_corr<22>"


def test_iq8_input_line():
    """`bench.py --input iq8`: the capture is an 8-bit IQ stream converted inside the forward transform; the satellites
    injected into the baseband capture are found after the engine's mixer, and the ingest stage is reported."""
    j = _bench("--input", "iq8", "--blocks-total", "256", "--no-e2e")
    assert "8-bit IQ" in j["config"]["input"] and j["config"]["cells_per_step_job"] == 256 * 73
    ing = j["ingest"]
    assert ing["kernel"] == "k_fwd2<iq8>" and ing["bytes_read"] == 256 * 80000 and 0 < ing["frac_of_copy_ceiling"] < 1
    assert j["roofline"]["traffic_stale"] is True  # the committed counters belong to the 1-bit line's kernel instance mix
    assert set(j["detected_prns"]) >= set(j["injected_prns_all_ranks"])


def test_default_line_carries_the_round3_fields():
    j = _bench("--blocks-total", "640", "--weak-blocks", "0")
    assert j["rccl_ranks_seen"] == 1 and j["blocks_per_rank"] == [640] and j["n_gpus"] == 1
    r = j["roofline"]
    assert isinstance(r["traffic_stale"], bool) and len(r["kernel_source_sha"]) == 16
    e = j["e2e_cli"]
    assert e["runs_reported"] == 20 and e["wall_s"] > 0 and e["hip_process_floor_s"] > 0 and "SearchTask" in e["split_ms"]


def test_two_self_spawned_ranks_share_the_capture():
    """`bench.py --gpus 2` launched plainly: it starts its own two ranks (gloo here, both on the one GPU), splits the runs between
    them and says so in the line."""
    j = _bench("--gpus", "2", "--blocks-total", "640", "--weak-blocks", "0", "--no-e2e", env={"GPSACQ_DIST_BACKEND": "gloo"})
    assert j["n_gpus"] == 2 and j["rccl_ranks_seen"] == 2 and j["dist_backend"] == "gloo"
    assert j["blocks_per_rank"] == [320, 320] and j["config"]["cells_per_step_job"] == 640 * 73
    assert set(j["detected_prns"]) >= set(j["injected_prns_all_ranks"])



def test_eight_ranks_share_the_nottingham_sized_capture():
    """The driver's N = 8 launch line (`python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 ... bench.py --gpus 8`) on the
    one-GPU box: gloo collectives, all eight ranks on device 0, the capture of the headline configuration (340 runs = 10 880
    blocks).  Pins what the first real SCALE run must not trip over: the split of 340 runs over 8 ranks (43 x 4 + 42 x 4 runs),
    eight ranks seen by the collective, the capture's satellites in the merged keys, the in-process multi-GPU leg (eight engines of the C
    ABI on the one GPU) reproducing those keys, and a stdout that carries the JSON line and nothing else (seven other ranks, the child
    process and torchrun write elsewhere)."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    e = dict(os.environ, GPSACQ_DIST_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "8", "--blocks-total", "10880", "--steps", "3",
                        "--no-cpu-baseline", "--no-e2e"], capture_output=True, text=True, timeout=900, env=e)
    assert r.returncode == 0, r.stderr[-3000:]
    out_lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(out_lines) == 1 and out_lines[0].startswith("{"), r.stdout[-2000:]
    j = json.loads(out_lines[0])
    assert j["n_gpus"] == 8 and j["rccl_ranks_seen"] == 8 and j["dist_backend"] == "gloo" and j["steps"] == 3
    assert j["blocks_per_rank"] == [1376] * 4 + [1344] * 4 and sum(j["blocks_per_rank"]) == 10880
    assert j["config"]["cells_per_step_job"] == 10880 * 73 and j["config"]["blocks_rank0"] == 1376 and j["scaling"] == "strong"
    assert len(j["injected_prns_all_ranks"]) == 8  # ONE capture (seed 1000), every rank its own blocks of it
    assert set(j["detected_prns"]) >= set(j["injected_prns_all_ranks"]) and len(j["keys_digest"]) == 16
    im = j["extras"]["inproc_multi"]
    assert "error" not in im and im["keys_equal_digest"] is True and im["runs"] == 340, im
    assert j["weak_scaling"]["blocks_per_gpu"] == 4096 and j["weak_scaling"]["value"] > 0
    assert j["value"] > 0 and 0 < j["roofline"]["frac"] < 1 and "cpu_baseline" not in j
