#!/bin/bash
# tools/exp_w52team.sh OUTDIR : 52-bit class at 2^15..2^17: the inverse (and the forward) per pass against the one-launch form at several lags
out=$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
S="timeout 300 python3 tools/sweep.py --qs 0xffffffff00001 --bytes 4e9 --steps 10"
(for n in 15 16 17; do
  echo "2^$n per pass: $($S --logn $n --ops fwd inv fwd inv --xcd-local 0 | tail -n +2 | awk '{printf "%s %s | ", $4, $8}')"
  for lag in 6 8 10 12 16 24; do
    echo "2^$n one launch lag $lag: $($S --logn $n --ops fwd inv fwd inv --xcd-local 1 --lag $lag | tail -n +2 | awk '{printf "%s %s | ", $4, $8}')"
  done
done) > $out/w52_xcd_local_lag.txt 2>&1
cat $out/w52_xcd_local_lag.txt
