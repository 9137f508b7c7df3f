#!/bin/bash
# tools/exp_oversub.sh OUTDIR : (1) does the stride-aware addressing cost the block kernels anything (previous build against this one,
# one workgroup per slot in both)?  (2) workgroups per resident slot (NTT_OPT_BLOCK_OVERSUB) for every 2^12-point block launch:
# whole polynomials, blocks below a column pass, the integer policy, the NTT-domain and coefficient-domain products
out=$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
Q=0x7fffffffe0001
S="python3 tools/sweep.py --bytes 8e9 --steps 10"
row() { tail -n +2 | awk '{printf "%s:%s ", $4, $8}'; }
(for rep in 1 2; do
  for lib in build/libntt_prev.so ""; do
    echo "rep $rep ${lib:-this build}: $(NTT_LIB=$lib timeout 300 $S --qs $Q --logn 10 12 13 14 --ops fwd inv mul --oversub 1 | tail -n +2 | awk '{printf "2^%s %s %s | ", $1, $4, $8}')"
    echo "rep $rep ${lib:-this build} 2^16: $(NTT_LIB=$lib timeout 300 $S --qs $Q --logn 16 --ops fwd inv mul --oversub 1 | row)"
  done
done) > $out/ab_stride_addressing.txt 2>&1
(for rep in 1 2; do
  for o in 1 2 4 8 16 32; do echo "rep $rep 2^12 oversub $o: $(timeout 200 $S --qs $Q --logn 12 --ops fwd inv mul --oversub $o | row)"; done
  for lg in 15 16; do for o in 1 2 4 8; do echo "rep $rep 2^$lg per-pass oversub $o: $(timeout 200 $S --qs $Q --logn $lg --ops fwd inv --xcd-local 0 --oversub $o | row)"; done; done
  for o in 1 2 4 8; do echo "rep $rep 2^16 52-bit oversub $o: $(timeout 200 $S --qs 0xffffffff00001 --logn 16 --ops fwd inv --oversub $o | row)"; done
  for o in 1 4 8 16; do echo "rep $rep 2^12 60-bit oversub $o: $(timeout 200 $S --qs 0xffffffffffc0001 --logn 12 --ops fwd inv --oversub $o | row)"; done
  for o in 1 4 8 16; do echo "rep $rep 2^12 52-bit oversub $o: $(timeout 200 $S --qs 0xffffffff00001 --logn 12 --ops fwd inv --oversub $o | row)"; done
done) > $out/oversub_sweep.txt 2>&1
(for o in 1 8; do echo "== oversub $o"; timeout 300 python3 tools/domain_bench.py --logn 12 16 --k 1 3 --steps 6 --oversub $o; done) > $out/oversub_domain.txt 2>&1
cat $out/ab_stride_addressing.txt $out/oversub_sweep.txt $out/oversub_domain.txt
