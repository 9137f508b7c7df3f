#!/bin/bash
out=gpurun_out/r06d; mkdir -p $out
for r in 1 2 3; do
  for l in optimized-number-theoretic-transform-implementations_amd/libntt_mi355x.so build/libntt_hoist.so; do
    echo "== $l (rep $r)"; NTT_LIB=$PWD/$l timeout 300 python tools/product_slab_bench.py 2>&1 | grep -v "^#"
  done
done > $out/ab_product_hoist.txt 2>&1
cat $out/ab_product_hoist.txt
