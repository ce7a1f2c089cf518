#!/bin/bash
# sample clocks / power while the soak loop runs
python tools/soak.py --size 16384 --windows 50 --reps 200 > gpurun_out/soak_clk.log 2>&1 &
SP=$!
sleep 45
for i in 1 2 3 4 5 6; do rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|mclk\|power\|fclk" | tr -s ' ' | head -8; echo ---; sleep 1.5; done
wait $SP
tail -1 gpurun_out/soak_clk.log
echo "=== idle"
rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|mclk\|power" | tr -s ' ' | head -6
