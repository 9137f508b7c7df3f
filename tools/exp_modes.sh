#!/bin/bash
# tools/exp_modes.sh OUTDIR : the two throughput modes of the small transforms (2^8..2^10): do they follow the allocation, the number of
# workgroups, the order in which the blocks are walked (build/libntt_perm.so: -DNTT_BLOCK_PERM=1, a multiplicative permutation of the
# workgroup-sized groups of blocks)?  Three processes per setting: the mode changes from process to process.
out=$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
(for lg in 8 9 10; do for rep in 1 2 3; do for lib in "" build/libntt_perm.so; do
  echo "== 2^$lg process $rep ${lib:-shipped order}"
  NTT_LIB=$lib timeout 300 python3 tools/small_size_modes.py --logn $lg --gib 6 --allocs 3 --max-grid 0 2048 8192 32768 131072 2>&1 | grep "grid cap\|pid"
done; done; done) > $out/small_size_modes.txt 2>&1
cat $out/small_size_modes.txt
