#!/bin/bash
cd $GRAFT_REPO_ROOT
for lg in 15 16 17; do
  for lag in 6 8 12; do
    echo "== xcd-local wpc 3 lag $lag"; timeout 60 python3 tools/sweep.py --logn $lg --ops inv --qs 0x80000001c0001 --bytes 16e9 --xcd-local 1 --lag $lag --wpc 3 | tail -n +2
  done
done
