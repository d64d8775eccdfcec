#!/bin/bash
# Round 4, first GPU call: the new tests, then the whole GPU suite, the fold A/B, the multi-engine enqueue table, the front-end
# warm-up A/B and one default bench line.  Everything lands in gpurun_out/r04a/.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04a
mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_gpu_round4.py -x -q -s > $OUT/pytest_round4.log 2>&1; echo "round4 rc $?" | tee -a $OUT/pytest_round4.log; tail -15 $OUT/pytest_round4.log
timeout 900 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "suite rc $?" | tee -a $OUT/pytest_gpu.log; tail -5 $OUT/pytest_gpu.log
timeout 600 python tools/ab_env.py GPSACQ_LIB=build/var_fold/libgpsacq.so GPSACQ_LIB=build/var_fold/libgpsacq.so > $OUT/ab_fold.log 2>&1; cat $OUT/ab_fold.log
timeout 300 python tools/multi_enqueue.py > $OUT/multi_enqueue.json 2> $OUT/multi_enqueue.err; cat $OUT/multi_enqueue.json
GPSACQ_MULTI_FORCE_RCCL=1 timeout 300 python tools/multi_enqueue.py > $OUT/multi_enqueue_rccl.json 2> $OUT/multi_enqueue_rccl.err; cat $OUT/multi_enqueue_rccl.json
timeout 300 python tools/e2e_warm.py > $OUT/e2e_warm.json 2> $OUT/e2e_warm.err; cat $OUT/e2e_warm.json
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; cut -c1-1500 $OUT/bench_default.json
tail -3 $OUT/*.err | cut -c1-300
