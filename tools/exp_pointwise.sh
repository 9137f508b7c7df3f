#!/bin/bash
# tools/exp_pointwise.sh OUTDIR : the pointwise product against the number of workgroups (NTT_OPT_MAX_GRID; 4194304 = one-shot workgroups,
# 0 = the library's rule: about four grid-stride iterations per workgroup), four operand sizes; copy shapes (tools/copy_variants.hip)
out=$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
(for b in 4e9 1e9 2.5e8 5e7; do echo "== operand bytes $b"; timeout 600 python3 tools/pointwise_bench.py $b sweep 2>&1 | grep "rep 1\|probe"; done) > $out/pointwise_grid.txt 2>&1
cat $out/pointwise_grid.txt
[ -x build/copy_variants ] && timeout 300 ./build/copy_variants > $out/copy_variants.txt 2>&1; cat $out/copy_variants.txt
