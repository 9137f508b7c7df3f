#!/bin/bash
# tools/ab_product_tail.sh: ntt_negacyclic_mul_batch on plans without the one-launch product kernel -- the last two steps as
# dot_inv_kernel (products inside the inverse's first pass) against pointwise + inverse launches (NTT_DOT_UNFUSED=1), same box,
# two alternating rounds: a 57-bit and a 60-bit modulus (integer policy), and an FP64 plan with the fused product switched off
cd $GRAFT_REPO_ROOT
for r in 1 2; do
  for u in 1 0; do
    echo "== NTT_DOT_UNFUSED=$u round $r"
    NTT_DOT_UNFUSED=$u python3 tools/sweep.py --logn 12 14 16 --ops mul --qs 0x1fffffffffc0001 0xffffffffffc0001 --bytes 2e9 2>&1 | grep -v "^logn"
    NTT_DOT_UNFUSED=$u python3 tools/sweep.py --logn 12 14 --ops mul --qs 0x7fffffffe0001 --fused-product 0 --bytes 2e9 2>&1 | grep -v "^logn"
  done
done
