#!/bin/bash
# Variant of libgpsacq.so that differs in acq_kernels.hip only (extra compiler flags), linked with the product's other objects:
#   tools/build_kvariant.sh <name> <flags...>   ->  build/var_<name>/libgpsacq.so   (run on the GPU box with GPSACQ_LIB=...)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
d=build/var_$name; mkdir -p $d
L=gnss-gps-sdr_amd/lib
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wno-unused-function "$@" -Rpass-analysis=kernel-resource-usage \
    -c gnss-gps-sdr_amd/csrc/acq_kernels.hip -o $d/k.o 2> $d/resources.txt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $d/libgpsacq.so $d/k.o $L/key_kernels.o $L/iq_kernels.o $L/gen_kernels.o $L/gpsacq_engine.o $L/gpsacq_multi.o -ldl -pthread
grep -E "Function Name|VGPRs:|ScratchSize|Occupancy" $d/resources.txt | sed -e 's/.*remark: [^ ]* *//' -e 's/\[-Rpass[^]]*\]//' | paste - - - - | grep "k_corrILi22ELi3ELi2ELb0ELb0ELb0" | sed 's/_ZN3acq6k_corrI//; s/EEEvNS_8CorrArgsE//' | cut -c1-160
echo built $d/libgpsacq.so
