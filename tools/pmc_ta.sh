#!/bin/bash
# texture-addresser / data-return busy counters for k_corr: tools/pmc_ta.sh <tag>
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmcta_$TAG
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o -E "\b(TA|TD|TCP)_[A-Z0-9_]+(sum|avr|max)?\b" | sort -u > $OUT/avail.txt
i=0
for set in "TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE" "TD_TD_BUSY_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum" "TA_BUFFER_WAVEFRONTS_sum TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  env "$@" timeout 120 rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -o p -- python $R/bench.py --steps 2 --warmup 1 --blocks-total 1024 --weak-blocks 0 --no-cpu-baseline --no-live-traffic > $OUT/log$i.txt 2>&1
done
python - <<PY
import csv,collections,glob
agg=collections.defaultdict(list)
for f in glob.glob("$OUT/p*/p_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_corr" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg): print("  %-36s %.5g" % (k, sum(agg[k])/len(agg[k])))
PY
find $OUT -name "*.db" -delete
