#!/bin/bash
# A/B of kernel variants (build/var_<name>/libgpsacq.so) against the product, each listed variant once per pass, two passes.
TAG=${1:-ab}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
ARGS=""
for v in "$@"; do ARGS="$ARGS GPSACQ_LIB=build/var_$v/libgpsacq.so"; done
timeout 2000 python tools/ab_env.py $ARGS $ARGS > $OUT/ab.log 2>&1; cat $OUT/ab.log | sed 's/"bit_exact": //; s/"max_rel_pwr": [0-9.e-]*, //; s/"argmax_mismatch": //' | cut -c1-200
