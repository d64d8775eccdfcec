#!/bin/bash
# Round 4, fourth GPU call: live PMC traffic inside bench.py, the full default line.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04d
mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_gpu_round4.py -q -k "live_traffic or rccl_single_rank_bench" > $OUT/pytest.log 2>&1; echo "rc $?"; tail -5 $OUT/pytest.log
( time python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err ) 2>&1 | tail -3; echo "stdout lines: $(wc -l < $OUT/bench_default.json)"
python - <<PY
import json
j = json.loads(open("$OUT/bench_default.json").read())
r = j["roofline"]
print("value %.4e ms/step %.3f frac %.4f" % (j["value"], j["ms_per_step"], r["frac"]))
print("traffic", r["traffic"], r["traffic_source"][:80], r["traffic_live"], "stale", r["traffic_stale"], "onchip_stale", r["onchip_counters_stale"])
print("soak", json.dumps(j.get("soak"))[:400])
print("dist", j["dist_backend"], j["rccl_ranks_seen"], j["dist_note"])
print("cpu", json.dumps(j.get("cpu_baseline"))[:500])
print("e2e", json.dumps(j.get("e2e_cli"))[:500])
PY
tail -3 $OUT/bench_default.err | cut -c1-200
