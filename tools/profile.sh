#!/bin/bash
# Runs on the GPU box (via gpurun): kernel trace + stats, then PMC passes in their own runs.
# Usage: tools/profile.sh <tag> [bench args...]
TAG=${1:-r01}; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline "$@" > $OUT/bench_traced.log 2>&1
rocprofv3 -L > $OUT/counters_avail.txt 2>&1
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" \
  "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
  "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD" \
  "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $OUT/pmc$i -o p -- python $R/bench.py --steps 2 --warmup 1 --blocks 1024 --no-cpu-baseline "$@" > $OUT/pmc$i.log 2>&1
done
find $OUT -name "*.db" -delete; find $OUT -type f | head -50 > $OUT/files.txt
du -sh $OUT
