#!/bin/bash
# tools/exp_teamdot_lag.sh OUTDIR : the one-launch NTT-domain products (team_dot_kernel), fine lag sweep per size, k = 1..4,
# 51-bit modulus; then 60-bit and the broadcast key at a few lags
out=$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
D="timeout 600 python3 tools/domain_bench.py --steps 6 --no-broadcast"
row() { grep "a^, b^" | awk '{printf "k=%s %s | ", $3, $NF}'; }
(
echo "== per chunk: 2^15 $($D --logn 15 --k 1 2 3 4 --xcd-local 0 | row)"
echo "== per chunk: 2^16 $($D --logn 16 --k 1 2 3 4 --xcd-local 0 | row)"
echo "== per chunk: 2^17 $($D --logn 17 --k 1 2 3 4 --xcd-local 0 | row)"
for lag in 16 20 24 28 32 40 48; do echo "2^15 lag $lag: $($D --logn 15 --k 1 2 3 4 --xcd-local 1 --lag $lag | row)"; done
for lag in 8 9 10 11 12 14 16; do echo "2^16 lag $lag: $($D --logn 16 --k 1 2 3 4 --xcd-local 1 --lag $lag | row)"; done
for lag in 4 5 6 7 8 10; do echo "2^17 lag $lag: $($D --logn 17 --k 1 2 3 4 --xcd-local 1 --lag $lag | row)"; done
) > $out/teamdot_lag_fine.txt 2>&1
cat $out/teamdot_lag_fine.txt
(
for n in 15 16 17; do echo "== 60-bit per chunk 2^$n: $($D --bits 60 --logn $n --k 1 3 --xcd-local 0 | row)"; done
for lag in 16 24 32 48; do echo "60-bit 2^15 lag $lag: $($D --bits 60 --logn 15 --k 1 3 --xcd-local 1 --lag $lag | row)"; done
for lag in 8 10 12 16; do echo "60-bit 2^16 lag $lag: $($D --bits 60 --logn 16 --k 1 3 --xcd-local 1 --lag $lag | row)"; done
for lag in 4 6 8; do echo "60-bit 2^17 lag $lag: $($D --bits 60 --logn 17 --k 1 3 --xcd-local 1 --lag $lag | row)"; done
B="timeout 600 python3 tools/domain_bench.py --steps 6"
brow() { grep "bcast" | awk '{printf "k=%s %s | ", $3, $NF}'; }
for n in 15 16 17; do echo "== broadcast key per chunk 2^$n: $($B --logn $n --k 1 3 --xcd-local 0 | brow)"; done
for lag in 16 24 32 48; do echo "bcast 2^15 lag $lag: $($B --logn 15 --k 1 3 --xcd-local 1 --lag $lag | brow)"; done
for lag in 8 10 12 16; do echo "bcast 2^16 lag $lag: $($B --logn 16 --k 1 3 --xcd-local 1 --lag $lag | brow)"; done
for lag in 4 6 8; do echo "bcast 2^17 lag $lag: $($B --logn 17 --k 1 3 --xcd-local 1 --lag $lag | brow)"; done
) > $out/teamdot_lag_other.txt 2>&1
cat $out/teamdot_lag_other.txt
