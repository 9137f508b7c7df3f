#!/bin/bash
# tools/run_bench_all.sh OUTDIR : bench lines of BASELINE configs 2, 3, 4, 5 (one GPU)
out=$1; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 900 python3 bench.py > $out/bench_config4.json 2> $out/bench_config4.err
timeout 900 python3 bench.py --config 2 > $out/bench_config2.json 2> $out/bench_config2.err
timeout 900 python3 bench.py --config 3 > $out/bench_config3.json 2> $out/bench_config3.err
timeout 900 python3 bench.py --config 5 --steps 10 --warmup 2 > $out/bench_config5.json 2> $out/bench_config5.err
tail -c 600 $out/*.err; for f in $out/bench_config?.json; do python3 -c "
import json,sys
d=json.load(open('$f')); r=d['roofline']; c=d.get('cpu_baseline',{})
print('$f', d['value'], d['unit'], 'frac %.3f min %.3f med %.3f'%(r['frac'], r.get('frac_at_min',0), r.get('frac_at_median',0)), 'cpu', c.get('value'), c.get('cores'), c.get('all_core_over_single_core'))
"; done
