#!/bin/bash
# Round-3 second pass: gpu tests, A/B of the kernel variants under build/var_*/ against the product (cells compared, in-kernel
# rate), in-kernel rates of every instance, the default bench line.  Usage: tools/gpu_r03b.sh <tag> [variant names...]
TAG=${1:-r03b}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 1700 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $OUT/pytest_gpu.log
tail -6 $OUT/pytest_gpu.log
ARGS=""
for v in "$@"; do ARGS="$ARGS GPSACQ_LIB=build/var_$v/libgpsacq.so"; done
timeout 1200 python tools/ab_env.py $ARGS > $OUT/ab.log 2>&1; cat $OUT/ab.log | cut -c1-260
timeout 300 python tools/rate_other_fs.py 2>&1 | grep -v amdgpu.ids > $OUT/rates.log; cat $OUT/rates.log
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python - <<PY
import json
j = json.loads([l for l in open("$OUT/bench_default.json") if l.startswith("{")][-1])
print("bench value %.4e ms/step %.2f kernel_ms %.2f frac %.4f" % (j["value"], j["ms_per_step"], j["roofline"]["kernel_ms"], j["roofline"]["frac"]))
print("e2e", json.dumps(j.get("e2e_cli"))[:900])
print("cpu all cores", json.dumps(j.get("cpu_baseline_all_cores"))[:500])
PY
GPSACQ_TRACE=1 gnss-gps-sdr_amd/bin/gps_test tests/golden/gps_sig_tmp.bin 2.046e6 8.184e6 5000 2>&1 >/dev/null | grep trace
