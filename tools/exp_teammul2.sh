#!/bin/bash
# tools/exp_teammul2.sh OUTDIR : where the automatic choice of the one-launch fwd_mul should start (batch sweep, both forms)
out=$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
D="timeout 600 python3 tools/domain_bench.py --k 1"
row() { grep "fwd(a)\." | awk '{printf "%s %s %s | ", $3, $4, $NF}'; }
(for n in 15 16 17; do for polys in 128 256 512 1024; do for x in 0 1; do
  bytes=$(python3 -c "print($polys * 8 * 2**$n)")
  echo "2^$n $polys polynomials xcd-local $x: $($D --logn $n --bytes $bytes --steps 20 --xcd-local $x | row)"
done; done; done) > $out/domain_bench_xcd_local_mul_batch.txt 2>&1
cat $out/domain_bench_xcd_local_mul_batch.txt
