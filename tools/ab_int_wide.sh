#!/bin/bash
# tools/ab_int_wide.sh [probe args]: the wide integer policy (ArithU64X, NTT_ARITH_AUTO plans of q >= 2^52) against the reference's
# Harvey butterflies (NTT_INT_WIDE=0: the library's switch for this A/B) on the same box, two alternating rounds; every line is
# checked against the oracle (first, middle, last polynomial) before it is timed
cd $GRAFT_REPO_ROOT
for r in 1 2; do
  for w in 0 1; do
    echo "== NTT_INT_WIDE=$w round $r"
    NTT_INT_WIDE=$w python3 tools/int_policy_probe.py --arith auto "$@" 2>&1 | grep "^2\^"
  done
done
