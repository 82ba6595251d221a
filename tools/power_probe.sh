#!/bin/bash
# usage: power_probe.sh <label> <command...> : runs the command in the background and samples socket power / sclk while it runs
label=$1; shift
"$@" > /tmp/pp_out.txt 2>&1 &
pid=$!
sleep 0.7
for i in 1 2 3 4; do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power \(W\)|sclk" | sed 's/^/    /' | tr '\n' ' '
  echo
  sleep 0.35
done
wait $pid
echo "== $label"; grep -v "^# dupl" /tmp/pp_out.txt | tail -3
