#!/bin/bash
# Round 4: the evidence round on the final tree (tools/gpu_round.sh r04b) + a wider fuzz sweep.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
bash tools/gpu_round.sh r04b
OUT=$R/gpurun_out/round_r04b
( timeout 1500 python tools/fuzz_gpu.py 70000 2500 > $OUT/fuzz_70000.log 2>&1; echo "fuzz rc $?" >> $OUT/fuzz_70000.log ); tail -2 $OUT/fuzz_70000.log
( FUZZ_CLI=1 timeout 600 python tools/fuzz_gpu.py 71000 150 > $OUT/fuzz_cli.log 2>&1; echo "fuzz cli rc $?" >> $OUT/fuzz_cli.log ); tail -2 $OUT/fuzz_cli.log
( FUZZ_PLUMBING=1 timeout 600 python tools/fuzz_gpu.py 72000 200 > $OUT/fuzz_plumb.log 2>&1; echo "fuzz plumbing rc $?" >> $OUT/fuzz_plumb.log ); tail -2 $OUT/fuzz_plumb.log
