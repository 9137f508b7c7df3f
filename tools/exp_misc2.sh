#!/bin/bash
# tools/exp_misc2.sh OUTDIR : (1) is the 2^12 forward transform slower on a 2 GiB slab (config 2) than on 8 GB because it is
# measured first (clocks still settling) or because of the slab?  the same operations in both orders, 20 and 200 steps;
# (2) the one-launch NTT-domain products at six and eight pairs; (3) the automatic choice against both forced forms around
# its threshold; (4) tools/exp_modes.sh (the two modes of the small transforms)
out=$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
S="timeout 300 python3 tools/sweep.py --logn 12 --qs 0x3ffffffffc001"
(for rep in 1 2; do for bytes in 2.147e9 8e9; do for steps in 20 200; do
  echo "rep $rep bytes $bytes steps $steps fwd inv fwd inv: $($S --bytes $bytes --steps $steps --ops fwd inv fwd inv | tail -n +2 | awk '{printf "%s %s | ", $4, $8}')"
  echo "rep $rep bytes $bytes steps $steps inv fwd inv fwd: $($S --bytes $bytes --steps $steps --ops inv fwd inv fwd | tail -n +2 | awk '{printf "%s %s | ", $4, $8}')"
done; done; done) > $out/order_2p12.txt 2>&1
cat $out/order_2p12.txt
D="timeout 600 python3 tools/domain_bench.py --steps 6"
row() { grep "a^, b^" | awk '{printf "k=%s %s | ", $3, $NF}'; }
(for n in 15 16 17; do for x in 0 1; do
  echo "2^$n 51-bit xcd-local $x: $($D --no-broadcast --logn $n --k 6 8 --bytes 2e9 --xcd-local $x | row)"
done; done
for n in 15 16 17; do for polys in 512 1024; do for x in 0 1 -1; do
  bytes=$(python3 -c "print($polys * 8 * 2**$n)")
  echo "2^$n $polys polynomials xcd-local $x: $($D --no-broadcast --logn $n --k 1 3 --bytes $bytes --steps 20 --xcd-local $x | row)"
done; done; done) > $out/domain_more.txt 2>&1
cat $out/domain_more.txt
bash tools/exp_modes.sh $out > $out/modes.log 2>&1; tail -60 $out/small_size_modes.txt
