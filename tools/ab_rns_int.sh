#!/bin/bash
# tools/ab_rns_int.sh: RNS products over limbs of the wide integer policy (57-bit primes), one launch chain over the set
# (NTT_RNS_LOOP=0: three launches) against one chain per prime (=1: four launches per prime), small per-limb batches
cd $GRAFT_REPO_ROOT
for cfg in "14 4 1" "14 16 1" "14 16 8" "14 16 64" "12 16 4" "16 4 2" "16 16 2"; do
  set -- $cfg
  for loop in 1 0; do
    NTT_RNS_LOOP=$loop python3 tools/pipeline_bench.py --logn $1 --limbs $2 --batch $3 --bits 57 --steps 20 2>&1 | grep "^N=" | sed 's/  [0-9.]* M limb-products.*//'
  done
done
