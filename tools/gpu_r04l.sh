#!/bin/bash
# radix-25 with its inner twiddles folded into the second-stage butterflies (product) against 16 complex multiplies (build/var_r25old)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
OUT=$R/gpurun_out/r04l
mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $OUT/pytest_gpu.log; tail -3 $OUT/pytest_gpu.log
bash tools/ab_bench.sh 3 r25old=build/var_r25old/libgpsacq.so new=product 2>&1 | tee $OUT/ab_config1.log
for c in "--config 2" "--config 3 --doppler-step 250" "--config 4 --doppler-step 50"; do
  echo "== $c" | tee -a $OUT/ab_configs.log
  AB_BENCH_ARGS="$c" bash tools/ab_bench.sh 2 r25old=build/var_r25old/libgpsacq.so new=product 2>&1 | tee -a $OUT/ab_configs.log
done
