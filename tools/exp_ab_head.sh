#!/bin/bash
# tools/exp_ab_head.sh OUTDIR : the committed HEAD library (tools/build_head.sh -> build/libntt_prev.so) against the working tree's, same box,
# alternating, one workgroup per resident slot in both (--oversub 1): does the stride-aware block addressing cost anything?  Then the
# default bench line of the working tree.
out=$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
Q=0x7fffffffe0001
S="python3 tools/sweep.py --bytes 8e9 --steps 10"
(for rep in 1 2 3; do
  for lib in build/libntt_prev.so ""; do
    echo "rep $rep ${lib:-this build}: $(NTT_LIB=$lib timeout 300 $S --qs $Q --logn 8 10 12 13 14 --ops fwd inv mul --oversub 1 | tail -n +2 | awk '{printf "2^%s %s %s | ", $1, $4, $8}')"
    echo "rep $rep ${lib:-this build} 2^16, 2^17: $(NTT_LIB=$lib timeout 300 $S --qs $Q --logn 16 17 --ops fwd inv mul --oversub 1 | tail -n +2 | awk '{printf "2^%s %s %s | ", $1, $4, $8}')"
    echo "rep $rep ${lib:-this build} 60-bit: $(NTT_LIB=$lib timeout 300 $S --qs 0xffffffffffc0001 --logn 12 14 16 --ops fwd inv --oversub 1 | tail -n +2 | awk '{printf "2^%s %s %s | ", $1, $4, $8}')"
  done
done) > $out/ab_head_vs_tree.txt 2>&1
cat $out/ab_head_vs_tree.txt
timeout 900 python3 bench.py > $out/bench_default.json 2> $out/bench_default.err
tail -c 6000 $out/bench_default.json
(for lm in "" "--batch-major"; do for r in 1 2; do timeout 300 python3 tools/pipeline_bench.py $lm; done; timeout 300 python3 tools/pipeline_bench.py --logn 14 --batch 4096 $lm; timeout 300 python3 tools/pipeline_bench.py --logn 14 --batch 2 --limbs 16 $lm; timeout 300 python3 tools/pipeline_bench.py --logn 16 --batch 1024 $lm; done) > $out/pipeline_layouts.txt 2>&1
cat $out/pipeline_layouts.txt
