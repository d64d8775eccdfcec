#!/bin/bash
# One GPU-box pass: gpu-marked tests, smoke, in-kernel rates of every instance, a short bench.  Usage: tools/gpu_check.sh <tag>
TAG=${1:-check}
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $OUT/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1
timeout 300 python tools/rate_other_fs.py > $OUT/rates.log 2>&1
timeout 600 python bench.py --steps 5 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err
tail -5 $OUT/pytest_gpu.log; tail -2 $OUT/smoke.log; cat $OUT/rates.log | grep -v amdgpu.ids; cut -c1-600 $OUT/bench.json; tail -3 $OUT/bench.err
GPSACQ_PROF=1 timeout 300 python bench.py --steps 1 --warmup 1 --weak-blocks 0 --no-cpu-baseline > $OUT/phase_profile.log 2>&1; grep "profile" $OUT/phase_profile.log | tail -2
