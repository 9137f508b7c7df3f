#!/bin/bash
# tools/exp_modes2.sh OUTDIR : the small transforms against the number of workgroups per launch (NTT_OPT_MAX_GRID), both directions,
# large and small batches, two processes per setting (the mode of an allocation changes from process to process)
out=$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
(for lg in 6 7 8 9 10 11; do for gib in 6 1; do for rep in 1 2; do
  echo "== 2^$lg $gib GiB process $rep"
  timeout 300 python3 tools/small_size_modes.py --logn $lg --gib $gib --allocs 2 --no-offsets --ops fwd inv --max-grid 0 16384 32768 65536 131072 262144 1048576 2>&1 | grep "grid cap"
done; done; done) > $out/small_size_grid.txt 2>&1
cat $out/small_size_grid.txt
