#!/bin/bash
# Samples board power / sclk with rocm-smi every 2 s while bench.py runs (diagnostic; needs a GPU box).
python bench.py --steps ${STEPS:-500} --warmup 2 --no-cpu-baseline > gpurun_out/power_bench.json 2> gpurun_out/power_bench.err &
BP=$!
n=0
while kill -0 $BP 2>/dev/null && [ $n -lt 120 ]; do
  rocm-smi --showpower --showclocks 2>&1 | grep -E "Package Power|sclk" | sed 's/.*: //' | tr '\n' ' '
  echo
  sleep 2
  n=$((n+1))
done
wait $BP
cut -c1-200 gpurun_out/power_bench.json
