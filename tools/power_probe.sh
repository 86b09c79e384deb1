#!/bin/bash
# Power cap, package power and shader clock while the reference-faithful loop runs (the perception convs dominate it).
rocm-smi --showmaxpower --showpower --showclocks 2>&1 | grep -i "power\|sclk\|mclk" | head -12
echo "--- under load (tools/faithful_only.py in the background) ---"
STEPS=6000 python3 tools/faithful_only.py > gpurun_out/power_probe_faithful.txt 2>&1 &
PID=$!
sleep 25
for i in 1 2 3 4 5 6; do
  rocm-smi --showpower --showclocks 2>&1 | grep -i "power\|sclk" | tr '\n' ' '; echo
  sleep 1
done
kill $PID 2>/dev/null; wait $PID 2>/dev/null || true
