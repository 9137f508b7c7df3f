#!/bin/bash
# tools/exp_warm.sh OUTDIR : config 2's line against the length of its warm-up (the clocks settle for ~20 ms after the idle gap of the
# parity check); then the default line
out=$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
pr() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); r=d['roofline']; print('value %.4g frac %.3f min-step frac %.3f median %.3f kernel_ms %.3f' % (d['value'], r['frac'], r.get('frac_at_min_step', 0), r.get('frac_at_median_step', 0), r['kernel_ms']))"; }
(for rep in 1 2; do for sw in "40 3" "40 10" "40 80" "100 80" "400 80"; do set -- $sw
  echo "rep $rep config 2 steps $1 warmup $2: $(timeout 600 python3 bench.py --config 2 --steps $1 --warmup $2 --no-cpu-baseline --headline-only | pr)"
done; done
for sw in "20 3" "20 8" "20 16"; do set -- $sw
  echo "config 4 steps $1 warmup $2: $(timeout 600 python3 bench.py --steps $1 --warmup $2 --no-cpu-baseline --headline-only | pr)"
done) > $out/warmup_length.txt 2>&1
cat $out/warmup_length.txt
timeout 900 python3 bench.py --no-cpu-baseline > $out/bench_default.json 2> $out/bench_default.err; python3 - <<PY
import json
d=json.loads(open("$out/bench_default.json").read().strip().split("\n")[-1])
print("headline", d["value"], d["roofline"]["frac"])
for k,v in d.items():
    if k.startswith("also_"):
        print(k, {kk: v.get(kk) for kk in ("value","frac","frac_at_min","steps","warmup","traffic_over_algorithmic","error")} if isinstance(v, dict) else v)
PY
