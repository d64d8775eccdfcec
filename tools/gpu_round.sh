#!/bin/bash
# Round evidence on the GPU box: gpu tests, bench lines of every configuration (default line with cpu baselines and the e2e leg),
# the IQ ingest lines, 2-rank runs of bench.py on the one GPU (gloo collectives; self-spawned and under torchrun), then the
# rocprofv3 trace + PMC passes of the default bench (tools/profile.sh).  Usage: tools/gpu_round.sh <tag>
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/round_$TAG
mkdir -p $OUT
cd $R
timeout 1700 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $OUT/pytest_gpu.log; tail -4 $OUT/pytest_gpu.log
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python bench.py --config 2 --bare > $OUT/bench_config2.json 2> $OUT/bench_config2.err
python bench.py --config 3 --doppler-step 250 --bare --steps 5 > $OUT/bench_config3.json 2> $OUT/bench_config3.err
python bench.py --config 4 --doppler-step 50 --bare --steps 5 > $OUT/bench_config4.json 2> $OUT/bench_config4.err
python bench.py --config 1 --input iq8 --steps 5 --bare > $OUT/bench_iq8_config1.json 2> $OUT/bench_iq8_config1.err
python bench.py --config 3 --input iq8 --blocks-total 1024 --steps 3 --bare > $OUT/bench_iq8_config3.json 2> $OUT/bench_iq8_config3.err
GPSACQ_DIST_BACKEND=gloo python bench.py --gpus 2 --steps 5 --warmup 2 --no-e2e 2> $OUT/bench_two_rank_selfspawn.err | grep '^{' > $OUT/bench_two_rank_selfspawn.json
GPSACQ_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
    bench.py --gpus 2 --steps 5 --warmup 2 --no-e2e 2> $OUT/bench_two_rank_gloo.err | grep '^{' > $OUT/bench_two_rank_gloo.json
GPSACQ_DIST_BACKEND=gloo python bench.py --gpus 2 --config 4 --doppler-step 50 --steps 3 --warmup 1 2> $OUT/bench_two_rank_gloo_config4.err | grep '^{' > $OUT/bench_two_rank_gloo_config4.json
# the driver's N = 8 launch line on the one GPU (gloo collectives, eight ranks on device 0): the Nottingham-sized capture split 43 x 4 + 42 x 4
# runs -- the SAME capture as the default line: keys_digest / detected must equal bench_default.json's; rank 0's share checked against the
# oracle; the in-process multi-GPU leg (eight engines of the C ABI) over the same capture
GPSACQ_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29519 \
    bench.py --gpus 8 --steps 5 --warmup 2 > $OUT/bench_eight_rank_gloo.json 2> $OUT/bench_eight_rank_gloo.err
python tools/multi_enqueue.py > $OUT/multi_enqueue.json 2> $OUT/multi_enqueue.err
GPSACQ_MULTI_FORCE_RCCL=1 python tools/multi_enqueue.py 2> $OUT/multi_enqueue_force_rccl.err | grep "^{" > $OUT/multi_enqueue_force_rccl.json  # (RCCL prints a version banner to stdout)
for i in 1 2 3; do GPSACQ_TRACE=1 gnss-gps-sdr_amd/bin/gps_test tests/golden/gps_sig_tmp.bin 2.046e6 8.184e6 5000 2>&1 >/dev/null | grep trace; done > $OUT/cli_trace.log; cat $OUT/cli_trace.log
bash tools/profile.sh $TAG
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/bench_*.json")):
    try:
        j = json.loads([l for l in open(f) if l.startswith("{")][-1])
        r = j["roofline"]
        print(f.split("/")[-1], "value %.4e ms/step %.2f kernel_ms %.3f frac %.4f ranks %s backend %s stale %s sample_ms %s" % (j["value"], j["ms_per_step"], j["roofline"]["kernel_ms"], j["roofline"]["frac"], j.get("rccl_ranks_seen"), j.get("dist_backend"), j["roofline"].get("traffic_stale"), (j.get("stage_ms") or {}).get("ms_sample")))
        print("    sclk %s MHz power %s W cycles/cell/CU %s frac_at_clock %s pk_fma_stream_TF %s frac_of_stream %s" % (r.get("sclk_mhz"), r.get("power_w"), r.get("cycles_per_cell_per_cu"), r.get("frac_at_clock"), r.get("pk_fma_stream_TF"), r.get("frac_of_pk_fma_stream")))
        if "cpu_baseline" in j: print("    parity:", json.dumps({k: v for k, v in j["cpu_baseline"].items() if k.startswith("parity_") and k != "parity_vs_gpu"}))
        print("    keys_digest", j.get("keys_digest"), "detected", j.get("detected_prns"), "parity_ok", j.get("parity_ok", (j.get("cpu_baseline") or {}).get("parity_ok")))
        print("    co-limiters: valu %s lds %s l2 %s uJ/cell %s" % (r.get("valu_busy_frac"), r.get("lds_frac"), r.get("l2_frac"), r.get("energy_uj_per_cell")))
        for k in ("gpu_library_baseline", "inproc_multi"):
            if k in j.get("extras", {}): print("   ", k, json.dumps(j["extras"][k])[:600])
        for k in ("soak", "strong_share_at_8", "one_rank_collective"):
            if k in j: print("   ", k, json.dumps(j[k])[:500])
        for k in ("e2e_cli", "ingest"):
            if k in j: print("   ", k, json.dumps(j[k])[:700])
    except Exception as ex:
        print(f, "unreadable:", ex)
PY
tail -n 3 $OUT/*.err | cut -c1-300
