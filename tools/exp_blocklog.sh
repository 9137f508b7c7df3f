#!/bin/bash
# tools/exp_blocklog.sh OUTDIR : 2^15 / 2^16 per-pass transforms on 2^12- against 2^14-point blocks (NTT_OPT_BLOCK_LOG), 51- and 52-bit moduli
out=$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
(for q in 0x7fffffffe0001 0xffffffff00001; do for bl in 12 14; do
  echo "q $q blocks 2^$bl: $(timeout 300 python3 tools/sweep.py --qs $q --logn 15 16 --ops fwd inv fwd inv --bytes 4e9 --steps 10 --xcd-local 0 --block-log $bl | tail -n +2 | awk '{printf "2^%s %s %s | ", $1, $4, $8}')"
done; done) > $out/block_log.txt 2>&1
cat $out/block_log.txt
