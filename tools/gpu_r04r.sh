#!/bin/bash
# final tree: smoke(), then a fuzz soak
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
OUT=$R/gpurun_out/r04r
mkdir -p $OUT
( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; echo "smoke rc $?" >> $OUT/smoke.log ); tail -3 $OUT/smoke.log
( timeout 420 python tools/fuzz_gpu.py 93000 2200 > $OUT/fuzz_general.log 2>&1; echo "fuzz general rc $?" >> $OUT/fuzz_general.log ); tail -2 $OUT/fuzz_general.log
( FUZZ_FS=7.9e6,8.3e6 timeout 120 python tools/fuzz_gpu.py 95500 400 > $OUT/fuzz_fs8.log 2>&1; echo "fuzz fs8 rc $?" >> $OUT/fuzz_fs8.log ); tail -2 $OUT/fuzz_fs8.log
( FUZZ_FS=2.6e6,3.0e6 timeout 120 python tools/fuzz_gpu.py 96000 400 > $OUT/fuzz_fs28.log 2>&1; echo "fuzz fs2.8 rc $?" >> $OUT/fuzz_fs28.log ); tail -2 $OUT/fuzz_fs28.log
