#!/bin/bash
# Round-3 first pass on the GPU box: all gpu tests (no -x: every failure is wanted), smoke, the default bench line with the
# e2e leg, the IQ ingest lines and the self-spawned 2-rank run (gloo on the one GPU).  Usage: tools/gpu_r03a.sh <tag>
TAG=${1:-r03a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 1700 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $OUT/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
timeout 600 python bench.py --config 1 --input iq8 --steps 5 --no-cpu-baseline --no-e2e > $OUT/bench_iq8_config1.json 2> $OUT/bench_iq8_config1.err
timeout 600 python bench.py --config 3 --input iq8 --blocks-total 1024 --steps 3 --no-cpu-baseline --no-e2e > $OUT/bench_iq8_config3.json 2> $OUT/bench_iq8_config3.err
GPSACQ_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 5 --warmup 2 --no-e2e > $OUT/bench_two_rank_selfspawn.json 2> $OUT/bench_two_rank_selfspawn.err
tail -15 $OUT/pytest_gpu.log; tail -2 $OUT/smoke.log
for f in $OUT/*.json; do echo "== $f"; cut -c1-700 $f; done
tail -3 $OUT/*.err | cut -c1-300
python - <<PY
import json
for n in ("bench_default", "bench_iq8_config1", "bench_iq8_config3", "bench_two_rank_selfspawn"):
    try:
        j = json.load(open("$OUT/%s.json" % n))
        print(n, "value %.3e" % j["value"], "ms/step %.2f" % j["ms_per_step"], "frac %.3f" % j["roofline"]["frac"], "ranks", j.get("rccl_ranks_seen"), j.get("blocks_per_rank"))
        for k in ("e2e_cli", "ingest", "cpu_baseline", "cpu_baseline_all_cores"):
            if k in j: print("  ", k, json.dumps(j[k])[:600])
    except Exception as ex:
        print(n, "unreadable:", ex)
PY
