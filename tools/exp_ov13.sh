#!/bin/bash
# tools/exp_ov13.sh OUTDIR : workgroups per resident slot at 2^13 (forward: 1024 threads x 2 blocks; inverse: 512 threads, two workgroups per CU)
out=$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
(for rep in 1 2; do for ov in 1 2 4 8 16; do
  echo "rep $rep oversub $ov: $(timeout 300 python3 tools/sweep.py --logn 13 --ops fwd inv fwd inv --bytes 8e9 --steps 10 --oversub $ov | tail -n +2 | awk '{printf "%s %s | ", $4, $8}')"
done; done) > $out/oversub_2p13.txt 2>&1
cat $out/oversub_2p13.txt
