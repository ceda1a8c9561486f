#!/bin/bash
# power / clock samples of the GPU while bench.py runs (is the step power-limited?)
python3 bench.py --steps 2500 --warmup 20 > gpurun_out/power_bench.log 2>&1 &
BP=$!
for i in $(seq 1 120); do
  rocm-smi --showpower --showclocks 2>&1 | grep -E "Package Power|sclk" | sed 's/.*: //' | tr '\n' ' '
  echo
  sleep 0.25
  kill -0 $BP 2>/dev/null || break
done | sort | uniq -c | sort -k1,1nr | head -40
wait $BP
tail -1 gpurun_out/power_bench.log | cut -c1-160
