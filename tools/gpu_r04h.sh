#!/bin/bash
# fold v2 (LDS-DMA table refresh) as a product candidate: parity suite + fuzz on the variant library, bench A/B on configs 1-4.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04h
mkdir -p $OUT
cd $R
V=$R/build/var_fold2/libgpsacq.so
GPSACQ_LIB=$V timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_extras.py tests/test_gpu_grid.py tests/test_gpu_fuzz.py tests/test_iq.py -m gpu -q > $OUT/pytest_variant.log 2>&1; echo "variant suite rc $?"; tail -3 $OUT/pytest_variant.log
( GPSACQ_LIB=$V timeout 600 python tools/fuzz_gpu.py 80000 500 > $OUT/fuzz_variant.log 2>&1; echo "fuzz rc $?" >> $OUT/fuzz_variant.log ); tail -2 $OUT/fuzz_variant.log
for cfg in "--config 1" "--config 2" "--config 3 --doppler-step 250" "--config 4 --doppler-step 50"; do
  for lib in product variant product variant; do
    if [ $lib = variant ]; then export GPSACQ_LIB=$V; else unset GPSACQ_LIB; fi
    python bench.py $cfg --steps 10 --warmup 2 --no-cpu-baseline --no-e2e --soak-seconds 0 --no-dist --weak-blocks 0 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$cfg', '$lib', 'value %.4e kernel_ms %.3f kcells %.3f M frac %.4f' % (j['value'], j['roofline']['kernel_ms'], j['roofline']['kernel_cells_per_s']/1e6, j['roofline']['frac']))"
  done
done
