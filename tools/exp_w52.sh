#!/bin/bash
# tools/exp_w52.sh OUTDIR : the 52-bit policy's inverse butterflies (difference left unreduced where both inputs are reduced sums) against
# the committed library (build/libntt_prev.so: tools/build_head.sh), same box, alternating; then the GPU test suite and config 3
out=$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
S="python3 tools/sweep.py --bytes 8e9 --steps 10 --qs 0xffffffff00001"
(for rep in 1 2 3; do
  for lib in build/libntt_prev.so ""; do
    echo "rep $rep ${lib:-this build}: $(NTT_LIB=$lib timeout 300 $S --logn 12 13 14 16 17 --ops fwd inv mul --oversub 1 | tail -n +2 | awk '{printf "2^%s %s %s | ", $1, $4, $8}')"
  done
done) > $out/ab_52bit_inverse.txt 2>&1
cat $out/ab_52bit_inverse.txt
timeout 2400 python3 -m pytest tests -m gpu -x -q > $out/pytest_gpu.txt 2>&1; tail -5 $out/pytest_gpu.txt
for r in 1 2; do timeout 600 python3 bench.py --config 3 --steps 20 --warmup 3 --no-cpu-baseline --headline-only | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('config 3: value %.4g frac %.3f kernel_ms %.3f' % (d['value'], d['roofline']['frac'], d['roofline']['kernel_ms']))"; done
(for r in 1 2; do for lm in "" "--batch-major"; do timeout 300 python3 tools/pipeline_bench.py $lm | cut -c1-120; timeout 300 python3 tools/pipeline_bench.py --logn 16 --batch 1024 $lm | cut -c1-120; timeout 300 python3 tools/pipeline_bench.py --logn 15 --batch 2048 $lm | cut -c1-120; done; done) > $out/pipeline_layouts.txt 2>&1
cat $out/pipeline_layouts.txt
(for r in 1 2; do for loop in 1 0; do for lg in 14 16; do NTT_RNS_LOOP=$loop timeout 120 python3 tools/pipeline_bench.py --logn $lg --limbs 16 --batch 2 --steps 10 | cut -c1-110; done; done; done) > $out/pipeline_small.txt 2>&1
cat $out/pipeline_small.txt
