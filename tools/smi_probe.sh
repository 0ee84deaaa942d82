#!/bin/bash
# sample clocks/power while the bench runs
python bench.py --steps 300 --warmup 5 --no-cpu-baseline --no-kernel-timing > gpurun_out/smi_bench.json 2>/dev/null &
BP=$!
sleep 20
for i in 1 2 3 4 5 6; do
  rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|Power|Temperature \(Sensor (junction|edge)" | tr -s ' ' | head -8
  echo ---
  sleep 2
done
wait $BP
tail -1 gpurun_out/smi_bench.json | cut -c1-200
echo "=== idle"
sleep 3
rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr -s ' '
rocm-smi --showmaxpower 2>/dev/null | grep -i "max" | head -3
