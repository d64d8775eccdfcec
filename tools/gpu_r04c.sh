#!/bin/bash
# Round 4, third GPU call: the whole GPU suite on the tree with k_fwd2 as the only 1-bit forward kernel, the experiment library's
# own checks (k_corr8, round 3's forward kernel), a fuzz sweep on the new forward kernel.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04c
mkdir -p $OUT
cd $R
timeout 1200 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "suite rc $?" | tee -a $OUT/pytest_gpu.log; tail -8 $OUT/pytest_gpu.log
timeout 300 python tools/check_corr8.py > $OUT/check_corr8.log 2>&1; echo "corr8 rc $?"; tail -3 $OUT/check_corr8.log
GPSACQ_LIB=$R/build/var_exp/libgpsacq.so timeout 300 python tools/ab_env.py GPSACQ_FWD1=1 > $OUT/ab_exp_fwd1.log 2>&1; cat $OUT/ab_exp_fwd1.log
( timeout 600 python tools/fuzz_gpu.py 50000 400 > $OUT/fuzz_50000.log 2>&1; echo "fuzz rc $?" >> $OUT/fuzz_50000.log ) ; tail -14 $OUT/fuzz_50000.log
( FUZZ_CLI=1 timeout 300 python tools/fuzz_gpu.py 51000 60 > $OUT/fuzz_cli.log 2>&1; echo "fuzz cli rc $?" >> $OUT/fuzz_cli.log ); tail -4 $OUT/fuzz_cli.log
( FUZZ_PLUMBING=1 timeout 300 python tools/fuzz_gpu.py 52000 60 > $OUT/fuzz_plumb.log 2>&1; echo "fuzz plumbing rc $?" >> $OUT/fuzz_plumb.log ); tail -4 $OUT/fuzz_plumb.log
