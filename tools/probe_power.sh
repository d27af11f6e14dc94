#!/bin/bash
# Socket power and clocks while a kernel loop runs: tools/probe_power.sh <seconds> -- <command...>
# (dev aid: tells a power-limited kernel from a stalled one; rocm-smi samples every 0.5 s)
secs=$1; shift; shift
"$@" > /tmp/probe_power_cmd.log 2>&1 &
pid=$!
sleep 3
for i in $(seq 1 $((secs * 2))); do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Socket Graphics Package Power|sclk clock level|mclk clock level" | tr '\n' ' ' | sed 's/GPU\[0\]\s*: //g'; echo
  sleep 0.5
done
wait $pid
tail -4 /tmp/probe_power_cmd.log
