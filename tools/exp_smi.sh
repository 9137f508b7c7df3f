#!/bin/bash
# tools/exp_smi.sh OUT CMD... : runs CMD in the background and samples rocm-smi power/clocks beside it
out=$1; shift
"$@" > $out.log 2>&1 &
pid=$!
: > $out.smi
while kill -0 $pid 2>/dev/null; do
  rocm-smi --showpower --showclocks --showuse 2>/dev/null | grep -E "Power|sclk|mclk|fclk|busy" | tr '\n' ';' >> $out.smi
  echo >> $out.smi
  sleep 0.25
done
wait $pid
