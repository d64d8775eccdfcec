#!/bin/bash
# A/B of kernel variants (build/var_*) + SQ counter passes of the product and of round 2's lane map + front-end traces.
TAG=${1:-r03c}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
ARGS=""
for v in "$@"; do ARGS="$ARGS GPSACQ_LIB=build/var_$v/libgpsacq.so"; done
timeout 1500 python tools/ab_env.py $ARGS > $OUT/ab.log 2>&1; cat $OUT/ab.log | cut -c1-260
timeout 600 bash tools/pmc_quick.sh ${TAG}_layc > $OUT/pmc_layc.log 2>&1; tail -22 $OUT/pmc_layc.log
timeout 600 bash tools/pmc_quick.sh ${TAG}_layb GPSACQ_LIB=$PWD/build/var_layb/libgpsacq.so > $OUT/pmc_layb.log 2>&1; tail -22 $OUT/pmc_layb.log
timeout 300 python tools/rate_other_fs.py 2>&1 | grep -v amdgpu.ids > $OUT/rates.log; cat $OUT/rates.log
python - > /tmp/nott.bin.log <<PY
import numpy as np
np.random.default_rng(0).integers(0, 256, 340 * 32 * 5120, dtype=np.uint8).tofile("/dev/shm/nott_size.bin")
PY
for i in 1 2 3; do /usr/bin/time -f "wall %e s" env GPSACQ_TRACE=1 gnss-gps-sdr_amd/bin/gps_test /dev/shm/nott_size.bin 4.092e6 5.456e6 5000 2>&1 >/dev/null | grep -E "trace|wall"; done
for i in 1 2 3; do /usr/bin/time -f "hip_floor wall %e s" gnss-gps-sdr_amd/bin/hip_floor 2>&1 | tail -1; done
