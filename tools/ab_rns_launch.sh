#!/bin/bash
# tools/ab_rns_launch.sh [rounds] : RNS products of large batches, N = 2^15..2^17 -- one XCD-local launch over all limbs
# (NTT_RNS_LOOP=0) against one launch per limb (NTT_RNS_LOOP=1), alternating, same box
rounds=${1:-4}
cd $GRAFT_REPO_ROOT
for r in $(seq $rounds); do
  for loop in 1 0; do
    NTT_RNS_LOOP=$loop python3 tools/pipeline_bench.py --steps 10 | cut -c1-110
    NTT_RNS_LOOP=$loop python3 tools/pipeline_bench.py --logn 16 --batch 1024 --steps 10 | cut -c1-110
    NTT_RNS_LOOP=$loop python3 tools/pipeline_bench.py --logn 15 --batch 2048 --steps 10 | cut -c1-110
  done
done
