#!/bin/bash
# tools/ab_xcd_f64_inv.sh: INVERSE transforms at N = 2^15..2^17 as items of one launch (--xcd-local 1) against one launch per pass
# (0), FP64 policies (51-bit, 52-bit, and at 2^17 a 50- and a 51-bit modulus), same box, two alternating rounds, 4 GiB slabs
cd $GRAFT_REPO_ROOT
for r in 1 2; do
  for x in 0 1; do
    echo "== --xcd-local $x round $r"
    python3 tools/sweep.py --logn 15 16 17 --ops inv --qs 0x7fffffffe0001 0xffffffff00001 --bytes 4e9 --xcd-local $x 2>&1 | grep -v "^logn"
    python3 tools/sweep.py --logn 17 --ops inv --qs 0x3ffffffb80001 0x7ffffff9c0001 --bytes 4e9 --xcd-local $x 2>&1 | grep -v "^logn"
  done
done
