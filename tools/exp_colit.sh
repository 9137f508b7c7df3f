#!/bin/bash
# tools/exp_colit.sh OUTDIR : the column passes with one-shot workgroups (shipped) against 2 and 4 grid-stride iterations per workgroup
# (build/libntt_colit2.so, libntt_colit4.so: -DNTT_COLUMN_ITERS), per-pass transforms at 2^15..2^17, 51-bit modulus
out=$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
(for rep in 1 2; do for lib in "" build/libntt_colit2.so build/libntt_colit4.so; do
  echo "rep $rep ${lib:-shipped (one-shot)}: $(NTT_LIB=$lib timeout 300 python3 tools/sweep.py --logn 15 16 17 --ops fwd inv fwd inv --bytes 4e9 --steps 10 --xcd-local 0 | tail -n +2 | awk '{printf "2^%s %s %s | ", $1, $4, $8}')"
done; done) > $out/column_iterations.txt 2>&1
cat $out/column_iterations.txt
