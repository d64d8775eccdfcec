#!/bin/bash
# Round 4, final tree (q = 0 peeled): evidence round + a fuzz sweep.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
bash tools/gpu_round.sh r04h
OUT=$R/gpurun_out/round_r04h
( FUZZ_FS=5.25e6,5.5e6 timeout 200 python tools/fuzz_gpu.py 90000 500 > $OUT/fuzz_fold.log 2>&1; echo "fuzz fold rc $?" >> $OUT/fuzz_fold.log ); tail -2 $OUT/fuzz_fold.log
( timeout 300 python tools/fuzz_gpu.py 91000 1200 > $OUT/fuzz_general.log 2>&1; echo "fuzz general rc $?" >> $OUT/fuzz_general.log ); tail -2 $OUT/fuzz_general.log
