#!/bin/bash
# sample power / clocks while a kernel loop runs
rocm-smi --showpower --showclocks --showmaxpower --showtemp 2>&1 | grep -v "^=\|^$" | head -40
echo "--- idle above; now under load"
python3 tools/ablation_table.py one product --frames 32 --reps 3000 > /tmp/one.log 2>&1 &
PID=$!
sleep 25
for i in 1 2 3; do rocm-smi --showpower --showclocks --showtemp 2>&1 | grep -i "power\|sclk\|mclk\|fclk\|temp" | head -12; echo; sleep 1; done
wait $PID
cat /tmp/one.log | tail -2
