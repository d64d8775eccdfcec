#!/bin/bash
# product with k_corr<22, FOLD>: whole GPU suite, A/B against the nofold variant (same box), fuzz.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04i
mkdir -p $OUT
cd $R
timeout 1200 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "suite rc $?"; tail -3 $OUT/pytest_gpu.log | head -2
python tools/ab_env.py GPSACQ_LIB=build/var_nofold/libgpsacq.so GPSACQ_LIB=build/var_nofold/libgpsacq.so > $OUT/ab_nofold.log 2>&1; cat $OUT/ab_nofold.log | cut -c1-230
for lib in product nofold product nofold; do
  if [ $lib = nofold ]; then export GPSACQ_LIB=$R/build/var_nofold/libgpsacq.so; else unset GPSACQ_LIB; fi
  python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-e2e --soak-seconds 0 --no-dist --weak-blocks 0 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib', 'value %.4e ms/step %.3f kernel_ms %.3f kcells %.3f M frac %.4f' % (j['value'], j['ms_per_step'], j['roofline']['kernel_ms'], j['roofline']['kernel_cells_per_s']/1e6, j['roofline']['frac']))"
done
unset GPSACQ_LIB
( timeout 900 python tools/fuzz_gpu.py 90000 1200 > $OUT/fuzz.log 2>&1; echo "fuzz rc $?" >> $OUT/fuzz.log ); tail -2 $OUT/fuzz.log
