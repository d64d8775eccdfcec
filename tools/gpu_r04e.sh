#!/bin/bash
# Round 4, fifth GPU call: the IQ mixer's fast sign path (iq8_bit): IQ tests, IQ fuzz modes, IQ bench lines.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04e
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_iq.py tests/test_gpu_pipeline.py tests/test_gpu_fuzz.py tests/test_bench_modes.py -m gpu -q > $OUT/pytest.log 2>&1; echo "rc $?"; tail -4 $OUT/pytest.log
( timeout 900 python tools/fuzz_gpu.py 60000 600 > $OUT/fuzz_60000.log 2>&1; echo "fuzz rc $?" >> $OUT/fuzz_60000.log ); tail -3 $OUT/fuzz_60000.log; grep -c "mode iq8\|mode pipe_iq" $OUT/fuzz_60000.log
python bench.py --config 1 --input iq8 --steps 5 --no-cpu-baseline --no-e2e --soak-seconds 0 > $OUT/bench_iq8_config1.json 2> $OUT/bench_iq8_config1.err
python bench.py --config 3 --input iq8 --blocks-total 1024 --steps 3 --no-cpu-baseline --no-e2e --soak-seconds 0 > $OUT/bench_iq8_config3.json 2> $OUT/bench_iq8_config3.err
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/bench_*.json")):
    j = json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split("/")[-1], "value %.4e ms/step %.3f" % (j["value"], j["ms_per_step"]), json.dumps(j.get("ingest"))[:260], j.get("detected_prns"), j.get("injected_prns_all_ranks"))
PY
python tools/e2e_iq.py 2>&1 | tail -5
