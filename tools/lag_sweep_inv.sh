#!/bin/bash
# tools/lag_sweep_inv.sh: the INVERSE as items of one launch, lag and residency sweep (FP64, 51-bit modulus), against the per-pass path
cd $GRAFT_REPO_ROOT
echo "== per pass"; python3 tools/sweep.py --logn 15 16 --ops inv --qs 0x7fffffffe0001 --bytes 4e9 --xcd-local 0 2>&1 | grep -v "^logn"
for lag in 1 2 3 4 6 8 12 16 24; do
  echo "== lag $lag"
  python3 tools/sweep.py --logn 15 16 --ops inv --qs 0x7fffffffe0001 --bytes 4e9 --xcd-local 1 --lag $lag 2>&1 | grep -v "^logn"
done
for wpc in 1 2 3; do
  echo "== wpc $wpc (default lag)"
  python3 tools/sweep.py --logn 15 16 --ops inv --qs 0x7fffffffe0001 --bytes 4e9 --xcd-local 1 --wpc $wpc 2>&1 | grep -v "^logn"
done
