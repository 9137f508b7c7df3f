#!/bin/bash
# tools/ab_int_wide_domain.sh: the NTT-domain product kernels with ArithU64X's stages (NTT_INT_WIDE=1) against the reference's
# butterflies (=0) for a 57-bit and a 60-bit modulus, same box, two alternating rounds (tools/domain_bench.py rows)
cd $GRAFT_REPO_ROOT
for r in 1 2; do
  for w in 0 1; do
    echo "== NTT_INT_WIDE=$w round $r"
    NTT_INT_WIDE=$w python3 tools/domain_bench.py --logn 12 14 --k 1 8 --bits 57 60 --bytes 2e9 2>&1 | grep -v "^$"
  done
done
