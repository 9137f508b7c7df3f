#!/bin/bash
# tools/block_sweep_int.sh: block size below the column pass (2^12 or 2^14) for the wide integer policy's per-pass transforms at 2^15, 2^16
cd $GRAFT_REPO_ROOT
for r in 1 2; do for bl in 12 14; do
  echo "== --block-log $bl round $r"
  python3 tools/sweep.py --logn 15 16 --ops fwd inv --qs 0x1fffffffffc0001 0xffffffffffc0001 --bytes 4e9 --xcd-local 0 --block-log $bl 2>&1 | grep -v "^logn"
done; done
