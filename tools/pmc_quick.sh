#!/bin/bash
# quick SQ counter passes for k_corr: tools/pmc_quick.sh <tag> [env assignments...]
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmcq_$TAG
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_BUSY_CYCLES" \
           "TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  env "$@" rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -o p -- python $R/bench.py --steps 2 --warmup 1 --blocks-total 1024 --weak-blocks 0 --no-cpu-baseline --no-live-traffic --no-e2e > $OUT/log$i.txt 2>&1
done
python - <<PY
import csv,collections,glob
agg=collections.defaultdict(list)
for f in glob.glob("$OUT/p*/p_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_corr" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"])); agg["dur_us"].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
d={k: sum(v)/len(v) for k,v in agg.items()}
print("$TAG")
for k in sorted(d): print("  %-24s %.5g" % (k, d[k]))
wc=d.get("SQ_WAVE_CYCLES",1)
print("  share of wave cycles: active %.2f  valu %.2f  wait_inst %.2f  wait_any %.2f" % (d["SQ_ACTIVE_INST_ANY"]/wc, d["SQ_ACTIVE_INST_VALU"]/wc, d["SQ_WAIT_INST_ANY"]/wc, d["SQ_WAIT_ANY"]/wc))
print("  per cell: valu %.0f lds %.0f vmem %.0f salu %.0f wave-instr; lds conflict share %.2f" % (d["SQ_INSTS_VALU"]/74752, d["SQ_INSTS_LDS"]/74752, d["SQ_INSTS_VMEM_RD"]/74752, d["SQ_INSTS_SALU"]/74752, d["SQ_LDS_BANK_CONFLICT"]/d["SQ_LDS_IDX_ACTIVE"]))
PY
