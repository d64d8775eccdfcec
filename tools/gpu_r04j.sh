#!/bin/bash
# Round 4, final tree: fuzz aimed at the 22-column coherent instance (k_corr<22, FOLD>: fs 5.25-5.5 MHz) + a general sweep.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
OUT=$R/gpurun_out/r04j
mkdir -p $OUT
( FUZZ_FS=5.25e6,5.5e6 timeout 420 python tools/fuzz_gpu.py 80000 900 > $OUT/fuzz_fold.log 2>&1; echo "fuzz fold rc $?" >> $OUT/fuzz_fold.log ); tail -2 $OUT/fuzz_fold.log
( timeout 200 python tools/fuzz_gpu.py 81000 300 > $OUT/fuzz_general.log 2>&1; echo "fuzz general rc $?" >> $OUT/fuzz_general.log ); tail -2 $OUT/fuzz_general.log
( FUZZ_FS=5.25e6,5.5e6 FUZZ_PLUMBING=1 timeout 120 python tools/fuzz_gpu.py 82000 60 > $OUT/fuzz_fold_plumb.log 2>&1; echo "fuzz plumbing rc $?" >> $OUT/fuzz_fold_plumb.log ); tail -2 $OUT/fuzz_fold_plumb.log
( FUZZ_FS=5.25e6,5.5e6 FUZZ_CLI=1 timeout 120 python tools/fuzz_gpu.py 83000 40 > $OUT/fuzz_fold_cli.log 2>&1; echo "fuzz cli rc $?" >> $OUT/fuzz_fold_cli.log ); tail -2 $OUT/fuzz_fold_cli.log
