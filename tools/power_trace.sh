#!/bin/bash
# on the GPU box: sample board power / sclk from sysfs-backed rocm-smi while the bench step runs back to back (is the clock the
# chip holds under the step a power cap, and how far below its cap does it sit?)  Output: gpurun_out/power.log
cd "$(dirname "$0")/.."
L=gpurun_out/power.log; : > $L
rocm-smi --showmaxpower --showpower --showclocks 2>&1 | grep -v "^$" >> $L
( timeout -k 5 120 python bench.py --no-cpu-baseline --no-train --no-surface --no-alt --steps 300 --warmup 10 2>/dev/null | cut -c1-200 >> $L ) &
BP=$!
sleep 25   # engine build + weights
for i in $(seq 1 12); do
  rocm-smi --showpower --showclocks 2>&1 | grep -E "Power|sclk|mclk" | tr '\n' ' ' >> $L; echo >> $L
  sleep 0.5
done
wait $BP
