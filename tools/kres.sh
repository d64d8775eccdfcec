#!/bin/bash
# compile acq_kernels.hip with --save-temps into build/isa and list VGPRs / scratch per kernel
cd "$(dirname "$0")/.." && mkdir -p build/isa && cd build/isa && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-function --save-temps -Rpass-analysis=kernel-resource-usage $HIPFLAGS_EXTRA -c ../../gnss-gps-sdr_amd/csrc/acq_kernels.hip -o acq.o 2>&1 | grep -E "error|Function Name|VGPRs:|ScratchSize" | sed -e 's/remark:.*Function Name: //' -e 's/\[-Rpass[^]]*\]//g' -e 's/remark: [^ ]* //' | paste - - - | grep -E "error|k_corr" 
