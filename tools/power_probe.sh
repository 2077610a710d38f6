#!/bin/bash
# sample clocks / power while the beam kernel runs (DM_SCORER selects the arithmetic)
python tools/split_probe.py dismember_amd/libdismember_hip.so 262144 > /tmp/pp.log 2>&1 &
PID=$!
sleep 14
for i in 1 2 3 4 5 6; do
  rocm-smi --showclocks --showpower --showuse 2>/dev/null | grep -E "sclk|Power|GPU use|mclk" | tr '\n' ';'; echo
  sleep 0.25
done
wait $PID
tail -1 /tmp/pp.log
