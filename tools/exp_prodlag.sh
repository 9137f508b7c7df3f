#!/bin/bash
# tools/exp_prodlag.sh OUTDIR : config 5's product launch (team_product_kernel, four primes, 2^17 x 512 pairs) and its 2^16 / 2^15 siblings
# against the lag between the three passes, on the shipped library (the defaults 8 / 14 / 20 date from round 3)
out=$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
(for cfg in "17 512 4 6 8 10 12 16" "16 1024 8 10 12 14 16 20" "15 2048 12 16 20 24 28 32"; do set -- $cfg; n=$1; b=$2; shift 2
  timeout 200 python3 tools/pipeline_bench.py --logn $n --batch $b --steps 10 > /dev/null 2>&1   # (settle the clocks)
  for lag in 0 $@; do echo "2^$n x $b lag $lag: $(timeout 200 python3 tools/pipeline_bench.py --logn $n --batch $b --steps 10 --lag $lag | awk '{print $5, $6, $7, $8, $9}')"; done
done) > $out/product_lag.txt 2>&1
cat $out/product_lag.txt
