#!/bin/bash
# quick SQ counter pass for k_corr: tools/pmc_quick.sh <tag> [env assignments...]
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmcq_$TAG
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
env "$@" rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS --output-format csv -d $OUT -o p -- python $R/bench.py --steps 2 --warmup 1 --blocks 1024 --no-cpu-baseline > $OUT/log.txt 2>&1
python - <<PY
import csv,collections
agg=collections.defaultdict(list)
for r in csv.DictReader(open("$OUT/p_counter_collection.csv")):
    if "k_corr" in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"])); agg["dur_us"].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3); agg["vgpr"].append(float(r["VGPR_Count"])+float(r["Accum_VGPR_Count"]))
print("$TAG", {k: round(sum(v)/len(v),1) for k,v in agg.items()})
PY
