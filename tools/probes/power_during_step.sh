#!/bin/bash
# Is the headline step power-limited?  Samples the socket power (hwmon power1_average / power1_input, uW), its cap and the shader clock
# (freq1_input, Hz) from sysfs every 0.5 s while bench.py runs 200 steps.
H=$(ls -d /sys/class/drm/card*/device/hwmon/hwmon* 2>/dev/null | head -1)
echo "hwmon: $H"; ls $H 2>/dev/null | tr '\n' ' '; echo
echo "cap: $(cat $H/power1_cap 2>/dev/null) default cap: $(cat $H/power1_cap_default 2>/dev/null) max: $(cat $H/power1_cap_max 2>/dev/null)"
(python3 bench.py --conv-precision bf16x3 --steps 200 --warmup 5 --no-cpu-baseline --no-kernel-timer --no-forward-only --no-h2d > /tmp/bench_power.log 2>&1; echo done > /tmp/bench_power.done) &
for i in $(seq 1 140); do
  [ -f /tmp/bench_power.done ] && break
  echo "t=$i power_uW=$(cat $H/power1_average 2>/dev/null || cat $H/power1_input 2>/dev/null) sclk_Hz=$(cat $H/freq1_input 2>/dev/null) mclk_Hz=$(cat $H/freq2_input 2>/dev/null)"
  sleep 0.5
done
wait
tail -1 /tmp/bench_power.log | cut -c1-200
