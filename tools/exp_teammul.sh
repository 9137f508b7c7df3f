#!/bin/bash
# tools/exp_teammul.sh OUTDIR : c^ = fwd(a) (.) b^ (+ c^) at N = 2^15..2^17 as ONE launch (team_mul_kernel, --xcd-local 1) against the
# per-chunk launches (--xcd-local 0); lag sweep; parity tests first
out=$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 1500 python3 -m pytest tests/test_gpu_parity.py -x -q -k "xcd_local_forward_transform_times or xcd_local_fwd_mul or forward_transform_times_a_transformed or key_switching" > $out/pytest_mul.txt 2>&1; tail -3 $out/pytest_mul.txt
D="timeout 600 python3 tools/domain_bench.py --steps 6 --k 1"
row() { grep "fwd(a)\." | awk '{printf "%s %s %s | ", $3, $4, $NF}'; }
(
for rep in 1 2; do for n in 15 16 17; do for x in 0 1; do
  echo "rep $rep 2^$n xcd-local $x: $($D --logn $n --xcd-local $x | row)"
done; done; done
for n in 15 16 17; do for lag in 6 8 10 12 16 24 32; do
  echo "2^$n lag $lag: $($D --logn $n --xcd-local 1 --lag $lag | row)"
done; done
for n in 15 16 17; do for x in 0 1; do
  echo "60-bit 2^$n xcd-local $x: $($D --bits 60 --logn $n --xcd-local $x | row)"
  echo "52-bit 2^$n xcd-local $x: $($D --bits 52 --logn $n --xcd-local $x | row)"
done; done
) > $out/domain_bench_xcd_local_mul.txt 2>&1
cat $out/domain_bench_xcd_local_mul.txt
