#!/bin/bash
# Round 4, second GPU call: k_fwd2 (the 1-bit forward kernel's second form) -- whole GPU suite on it, A/B against round 3's kernel
# (GPSACQ_FWD1=1), bench lines with stdout checked to be the JSON line alone.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04b
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; echo "suite rc $?" | tee -a $OUT/pytest_gpu.log; tail -5 $OUT/pytest_gpu.log
timeout 600 python tools/ab_env.py GPSACQ_FWD1=1 GPSACQ_FWD1=1 > $OUT/ab_fwd.log 2>&1; cat $OUT/ab_fwd.log
python bench.py --steps 20 --no-cpu-baseline --no-e2e --soak-seconds 2 > $OUT/bench_new.json 2> $OUT/bench_new.err; echo "stdout lines: $(wc -l < $OUT/bench_new.json)"
GPSACQ_FWD1=1 python bench.py --steps 20 --no-cpu-baseline --no-e2e --soak-seconds 2 > $OUT/bench_fwd1.json 2> $OUT/bench_fwd1.err
python bench.py --config 4 --doppler-step 50 --no-cpu-baseline --steps 5 --soak-seconds 0 > $OUT/bench_c4_new.json 2> $OUT/bench_c4_new.err
GPSACQ_FWD1=1 python bench.py --config 4 --doppler-step 50 --no-cpu-baseline --steps 5 --soak-seconds 0 > $OUT/bench_c4_fwd1.json 2> $OUT/bench_c4_fwd1.err
python bench.py --config 1 --input iq8 --steps 5 --no-cpu-baseline --no-e2e --soak-seconds 0 > $OUT/bench_iq8_new.json 2> $OUT/bench_iq8_new.err
GPSACQ_FWD1=1 python bench.py --config 1 --input iq8 --steps 5 --no-cpu-baseline --no-e2e --soak-seconds 0 > $OUT/bench_iq8_fwd1.json 2> $OUT/bench_iq8_fwd1.err
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/bench_*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], "value %.4e ms/step %.3f kernel_ms %.3f frac %.4f stage %s soak %s ingest %s" % (j["value"], j["ms_per_step"], j["roofline"]["kernel_ms"], j["roofline"]["frac"], j.get("stage_ms"), json.dumps(j.get("soak"))[:300], json.dumps(j.get("ingest"))[:200]))
    except Exception as ex:
        print(f, "unreadable:", ex)
PY
for f in $OUT/*.err; do echo "== $f"; tail -n 3 $f | cut -c1-300; done
