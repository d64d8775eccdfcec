#!/bin/bash
# Runs on the GPU box (via gpurun): ONE workload -- bench.py at its default batch size (the whole 10 880-block capture per
# launch, weak leg off so that every k_corr launch has the same size) -- first under rocprofv3 --kernel-trace --stats, then
# under the PMC passes in their own runs (no trace domains together with --pmc).  tools/summarize_prof.py turns the
# directory into profiles/<tag>_summary.md, profiles/<tag>_kernel_stats.csv and profiles/traffic.json.
# Usage: tools/profile.sh <tag> [extra bench args...]
TAG=${1:-r02}; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 20 --warmup 5 --bare --no-dist $@"
(cd $R && python -c "import bench; print(bench.kernel_source_sha())") > $OUT/kernel_source_sha.txt  # what the counters belong to
echo "bench.py $ARGS" > $OUT/command.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python $R/bench.py $ARGS > $OUT/bench_traced.log 2>&1
python $R/bench.py $ARGS > $OUT/bench_unprofiled.log 2>&1
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" \
  "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
  "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD" \
  "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum" "TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $OUT/pmc$i -o p -- python $R/bench.py --steps 2 --warmup 1 --bare --no-dist "$@" > $OUT/pmc$i.log 2>&1
done
find $OUT -name "*.db" -delete; find $OUT -type f | head -50 > $OUT/files.txt
du -sh $OUT
