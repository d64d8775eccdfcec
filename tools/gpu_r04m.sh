#!/bin/bash
# Round 4, final tree (five-point butterfly in 15 instructions, radix-25 with folded inner twiddles): evidence round + fuzz.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
bash tools/gpu_round.sh r04g
OUT=$R/gpurun_out/round_r04g
( FUZZ_FS=5.25e6,5.5e6 timeout 300 python tools/fuzz_gpu.py 84000 1200 > $OUT/fuzz_fold.log 2>&1; echo "fuzz fold rc $?" >> $OUT/fuzz_fold.log ); tail -2 $OUT/fuzz_fold.log
( timeout 400 python tools/fuzz_gpu.py 86000 2000 > $OUT/fuzz_general.log 2>&1; echo "fuzz general rc $?" >> $OUT/fuzz_general.log ); tail -2 $OUT/fuzz_general.log
( FUZZ_PLUMBING=1 timeout 120 python tools/fuzz_gpu.py 88000 100 > $OUT/fuzz_plumb.log 2>&1; echo "fuzz plumbing rc $?" >> $OUT/fuzz_plumb.log ); tail -2 $OUT/fuzz_plumb.log
( FUZZ_CLI=1 timeout 120 python tools/fuzz_gpu.py 89000 100 > $OUT/fuzz_cli.log 2>&1; echo "fuzz cli rc $?" >> $OUT/fuzz_cli.log ); tail -2 $OUT/fuzz_cli.log
