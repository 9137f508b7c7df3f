#!/bin/bash
# tools/exp_dotnt.sh OUTDIR : team_dot_kernel's operand loads plain (shipped) against non-temporal (build/libntt_dotnt1.so: a and a
# per-polynomial b; dotnt2: a only), alternating, three operand kinds
out=$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
D="timeout 600 python3 tools/domain_bench.py --steps 6 --xcd-local 1"
row() { grep "a^, b^" | awk '{printf "k=%s %s | ", $3, $NF}'; }
brow() { grep "bcast" | awk '{printf "k=%s %s | ", $3, $NF}'; }
(for rep in 1 2; do for n in 15 16 17; do for lib in "" build/libntt_dotnt1.so build/libntt_dotnt2.so; do
  echo "rep $rep 2^$n ${lib:-shipped (plain)}: a^,b^ $(NTT_LIB=$lib $D --logn $n --k 1 2 3 8 | tee /tmp/d.txt | row) key: $(cat /tmp/d.txt | brow)"
done; done; done
for n in 15 16 17; do for lib in "" build/libntt_dotnt1.so build/libntt_dotnt2.so; do
  echo "60-bit 2^$n ${lib:-shipped (plain)}: $(NTT_LIB=$lib $D --bits 60 --no-broadcast --logn $n --k 1 3 | row)"
done; done) > $out/ab_teamdot_nt.txt 2>&1
cat $out/ab_teamdot_nt.txt
