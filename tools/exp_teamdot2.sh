#!/bin/bash
# tools/exp_teamdot2.sh OUTDIR : the one-launch NTT-domain products with the shipped lag defaults against the per-chunk launches,
# alternating, three operand kinds; where the automatic choice should start (batch sweep); then the GPU tests
out=$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 1500 python3 -m pytest tests/test_gpu_parity.py -x -q -k "xcd_local_ntt_domain or inv_dot or automatic_form or hip_graph" > $out/pytest_dot.txt 2>&1; tail -3 $out/pytest_dot.txt
D="timeout 600 python3 tools/domain_bench.py --steps 6"
row() { grep "a^, b^" | awk '{printf "k=%s %s | ", $3, $NF}'; }
brow() { grep "bcast" | awk '{printf "k=%s %s | ", $3, $NF}'; }
(
for rep in 1 2; do for n in 15 16 17; do for x in 0 1; do
  echo "rep $rep 2^$n 51-bit xcd-local $x: $($D --no-broadcast --logn $n --k 1 2 3 4 --xcd-local $x | row)"
done; done; done
for n in 15 16 17; do for x in 0 1; do
  echo "2^$n 60-bit xcd-local $x: $($D --no-broadcast --bits 60 --logn $n --k 1 3 --xcd-local $x | row)"
  echo "2^$n 52-bit xcd-local $x: $($D --no-broadcast --bits 52 --logn $n --k 1 3 --xcd-local $x | row)"
  echo "2^$n broadcast key xcd-local $x: $($D --logn $n --k 1 3 8 --xcd-local $x | brow)"
done; done
for lag in 10 12; do echo "2^17 broadcast key lag $lag: $($D --logn 17 --k 1 3 --xcd-local 1 --lag $lag | brow)"; done
) > $out/domain_bench_xcd_local.txt 2>&1
cat $out/domain_bench_xcd_local.txt
(
for n in 15 16 17; do for polys in 128 256 512 1024 2048; do for x in 0 1; do
  bytes=$(python3 -c "print($polys * 8 * 2**$n)")
  echo "2^$n $polys polynomials xcd-local $x: $($D --no-broadcast --logn $n --k 1 3 --bytes $bytes --steps 20 --xcd-local $x | row)"
done; done; done
) > $out/domain_bench_xcd_local_batch.txt 2>&1
cat $out/domain_bench_xcd_local_batch.txt
timeout 2400 python3 -m pytest tests -m gpu -x -q > $out/pytest_gpu.txt 2>&1; tail -5 $out/pytest_gpu.txt
