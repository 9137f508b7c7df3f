#!/bin/bash
# tools/ab_sweep.sh "<sweep.py args>" [rounds] : same-box A/B of build/libntt_prev.so (A) against the working tree's library (B)
args=$1; rounds=${2:-3}
cd $GRAFT_REPO_ROOT
for r in $(seq $rounds); do
  echo "== A (previous) round $r"; NTT_LIB=build/libntt_prev.so python3 tools/sweep.py $args | tail -n +2
  echo "== B (this tree) round $r"; python3 tools/sweep.py $args | tail -n +2
done
