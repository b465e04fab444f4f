#!/bin/bash
# Samples rocm-smi power / clocks while a command runs: tools/power_sample.sh OUT -- cmd args...
out=$1; shift; shift
"$@" > $out.cmd.log 2>&1 &
pid=$!
: > $out
while kill -0 $pid 2>/dev/null; do
  /opt/rocm/bin/rocm-smi --showpower --showclocks --showuse 2>/dev/null | grep -E "Power|sclk|mclk|fclk|busy" | tr '\n' '|' >> $out
  echo >> $out
  sleep 0.3
done
wait $pid
