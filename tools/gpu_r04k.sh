#!/bin/bash
# dft5 in 15 packed instructions (product) against the 18-instruction form (build/var_dft5old): GPU suite, interleaved bench A/B.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
OUT=$R/gpurun_out/r04k
mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $OUT/pytest_gpu.log; tail -3 $OUT/pytest_gpu.log
bash tools/ab_bench.sh 3 old=build/var_dft5old/libgpsacq.so new=product 2>&1 | tee $OUT/ab_config1.log
for c in "--config 2" "--config 3 --doppler-step 250" "--config 4 --doppler-step 50"; do
  echo "== $c" | tee -a $OUT/ab_configs.log
  AB_BENCH_ARGS="$c" bash tools/ab_bench.sh 2 old=build/var_dft5old/libgpsacq.so new=product 2>&1 | tee -a $OUT/ab_configs.log
done
