#!/bin/bash
out=$1; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 600 python3 -m pytest tests -m gpu -x -q -k "xcd_local" > $out/pytest_team.txt 2>&1; tail -3 $out/pytest_team.txt
for lg in 15 16 17; do
  echo "== per-pass"; timeout 60 python3 tools/sweep.py --logn $lg --ops fwd inv --qs 0x80000001c0001 --bytes 16e9 --xcd-local 0 | tail -n +2
  for wpc in 3; do for lag in 4 6 8 12 16; do
    echo "== xcd-local wpc $wpc lag $lag"; timeout 60 python3 tools/sweep.py --logn $lg --ops fwd inv --qs 0x80000001c0001 --bytes 16e9 --xcd-local 1 --lag $lag --wpc $wpc | tail -n +2
  done; done
done > $out/sweep_team.txt 2>&1
cat $out/sweep_team.txt
