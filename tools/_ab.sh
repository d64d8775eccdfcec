cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06e
O=gpurun_out/r06e/persist_ab_box_$1.log
: > $O
for i in 1 2 3; do
  for v in 0 1; do
    GPSACQ_CORR_PERSIST=$v python bench.py --bare --no-dist --steps 20 --warmup 5 2>/dev/null | python -c "
import sys, json
j = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r = j['roofline']; cs = r.get('clock_sampling') or {}
print('config1 persist=$v kernel_ms %.3f ms/step %.3f value %.4e frac %.4f sclk %.0f xcd %s' % (r['kernel_ms'], j['ms_per_step'], j['value'], r['frac'], r['sclk_mhz'] or 0, cs.get('sclk_mhz_per_xcd')))
" >> $O 2>&1
  done
done
cat $O
