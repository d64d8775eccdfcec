#!/bin/bash
# tools/power_probe.sh -- board power, clocks and temperature sampled every 0.5 s while the default bench leg runs
# (is k_corr's ~2.16 GHz the power cap or a fixed DVFS state?).  Output: gpurun_out/power_samples.txt
mkdir -p gpurun_out
( python bench.py --steps 200 --warmup 2 --no-e2e --no-cpu-baseline --no-live-traffic 2>/dev/null | grep '^{' | cut -c1-220 > gpurun_out/power_bench.txt ) &
BP=$!
: > gpurun_out/power_samples.txt
while kill -0 $BP 2>/dev/null; do
  echo "$(date +%s.%N | cut -c1-14) $(rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E 'Power|sclk|mclk|junction|hotspot' | sed -e 's/GPU\[0\]//' -e 's/ *: */:/g' | tr -s ' \t' ' ' | tr '\n' ';')" >> gpurun_out/power_samples.txt
  sleep 0.3
done
rocm-smi --showmaxpower 2>/dev/null | grep -i "max" >> gpurun_out/power_samples.txt
cat gpurun_out/power_bench.txt
sort -t: -k3 -n gpurun_out/power_samples.txt | tail -30
