#!/bin/bash
# samples the shader clock and the socket power with rocm-smi while a command runs: bash tools/probe/clock_watch.sh <command ...>
"$@" > /tmp/cw_out.txt 2>&1 &
pid=$!
sleep 1.0
for i in $(seq 1 14); do
  /opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|Power (W)" | sed 's/^GPU\[0\]\s*: //' | tr '\n' '|'
  echo
  sleep 0.3
  kill -0 $pid 2>/dev/null || break
done
wait $pid
tail -c 300 /tmp/cw_out.txt
