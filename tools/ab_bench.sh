#!/bin/bash
# Interleaved bench A/B of library variants on one box: tools/ab_bench.sh <rounds> <name=path|product> ...   (kernel ms of configs[1])
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
rounds=$1; shift
for r in $(seq 1 $rounds); do
  for v in "$@"; do
    name=${v%%=*}; path=${v#*=}
    if [ "$path" = product ]; then unset GPSACQ_LIB; else export GPSACQ_LIB=$R/$path; fi
    python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-live-traffic --no-e2e --soak-seconds 0 --no-dist --weak-blocks 0 ${AB_BENCH_ARGS} 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$name', 'kernel_ms %.3f ms/step %.3f kcells %.3f M frac %.4f' % (j['roofline']['kernel_ms'], j['ms_per_step'], j['roofline']['kernel_cells_per_s']/1e6, j['roofline']['frac']))"
  done
done
