#!/bin/bash
# Round 4: smoke(), the default bench line with the strong-share leg, the round-4 tests.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04g
mkdir -p $OUT
cd $R
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout 600 python -m pytest tests/test_gpu_round4.py -q > $OUT/pytest.log 2>&1; echo "rc $?"; tail -3 $OUT/pytest.log
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "stdout lines: $(wc -l < $OUT/bench_default.json)"
python - <<PY
import json
j = json.loads(open("$OUT/bench_default.json").read())
print("value %.4e ms/step %.3f frac %.4f" % (j["value"], j["ms_per_step"], j["roofline"]["frac"]))
print("share", json.dumps(j.get("strong_share_at_8")))
print("soak", json.dumps(j.get("soak"))[:300])
PY
