#!/bin/bash
# Timing-only ablation builds of libgpsacq.so (ACQ_ABL in acq_math.hpp): build/abl/libgpsacq_abl<N>.so.
# WRONG results by construction; used with GPSACQ_LIB=<path> GPSACQ_KVAR=2 tools/kvar_exp.py to see which pipe a pass waits for.
set -e
cd "$(dirname "$0")/.."
mkdir -p build/abl
for n in "$@"; do
  d=build/abl/o$n; mkdir -p $d
  F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -DACQ_ABL=$n"
  /opt/rocm/bin/hipcc $F -ffp-contract=off -c gnss-gps-sdr_amd/csrc/gpsacq_engine.cpp -o $d/e.o
  /opt/rocm/bin/hipcc $F -c gnss-gps-sdr_amd/csrc/gpsacq_multi.cpp -o $d/m.o
  /opt/rocm/bin/hipcc $F -c gnss-gps-sdr_amd/csrc/acq_kernels.hip -o $d/k.o
  /opt/rocm/bin/hipcc $F -c gnss-gps-sdr_amd/csrc/iq_kernels.hip -o $d/i.o
  /opt/rocm/bin/hipcc $F -c gnss-gps-sdr_amd/csrc/gen_kernels.hip -o $d/g.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/abl/libgpsacq_abl$n.so $d/k.o $d/i.o $d/g.o $d/e.o $d/m.o -ldl
  echo built build/abl/libgpsacq_abl$n.so
done
