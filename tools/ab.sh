#!/bin/bash
# A/B helper for the GPU box: runs bench.py under several env settings and prints one short line each.
# usage: tools/ab.sh "label1|ENV=.. ENV2=.." "label2|..."
for spec in "$@"; do
  label="${spec%%|*}"; envs="${spec#*|}"
  out=$(env $envs python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline $BENCH_ARGS 2>&1 | tail -1)
  echo "$label :: $(echo "$out" | python3 -c 'import sys,json
try:
    d=json.loads(sys.stdin.read()); print("%.2f M NTT/s  %.0f GB/s  frac %.3f  kernel_ms %.3f" % (d["value"]/1e6, d["roofline"]["achieved"], d["roofline"]["frac"], d["roofline"]["kernel_ms"]))
except Exception as e: print("FAILED", e)')"
done
