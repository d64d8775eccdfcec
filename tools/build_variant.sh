#!/bin/bash
# Build a second copy of libgpsacq.so with extra compiler flags (e.g. -DACQ_EXP_FOO) into build/var_<name>/ for A/B runs:
#   tools/build_variant.sh foo -DACQ_EXP_FOO && GPU box: python tools/ab_env.py GPSACQ_LIB=build/var_foo/libgpsacq.so
set -e
cd "$(dirname "$0")/.."
name=$1; shift
d=build/var_$name; mkdir -p $d
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function $@"
/opt/rocm/bin/hipcc $F -ffp-contract=off -c gnss-gps-sdr_amd/csrc/gpsacq_engine.cpp -o $d/e.o
/opt/rocm/bin/hipcc $F -c gnss-gps-sdr_amd/csrc/gpsacq_multi.cpp -o $d/m.o
/opt/rocm/bin/hipcc $F -c gnss-gps-sdr_amd/csrc/acq_kernels.hip -o $d/k.o
/opt/rocm/bin/hipcc $F -c gnss-gps-sdr_amd/csrc/key_kernels.hip -o $d/y.o
/opt/rocm/bin/hipcc $F -c gnss-gps-sdr_amd/csrc/iq_kernels.hip -o $d/i.o
/opt/rocm/bin/hipcc $F -c gnss-gps-sdr_amd/csrc/gen_kernels.hip -o $d/g.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $d/libgpsacq.so $d/k.o $d/y.o $d/i.o $d/g.o $d/e.o $d/m.o -ldl -pthread
echo built $d/libgpsacq.so
