#!/bin/bash
# tools/ab_xcd_int.sh: N = 2^15..2^17 transforms of the wide integer policy, both passes as items of one launch (--xcd-local 1)
# against one launch per pass and chunk (0), same box, two alternating rounds, 57- and 60-bit moduli, 4 GiB slabs
cd $GRAFT_REPO_ROOT
for r in 1 2; do
  for x in 0 1; do
    echo "== --xcd-local $x round $r"
    python3 tools/sweep.py --logn 15 16 17 --ops fwd inv --qs 0x1fffffffffc0001 0xffffffffffc0001 --bytes 4e9 --xcd-local $x 2>&1 | grep -v "^logn"
  done
done
