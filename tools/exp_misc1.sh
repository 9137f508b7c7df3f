#!/bin/bash
out=$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests/test_gpu_layouts.py -x -q > $out/pytest_layouts.txt 2>&1; tail -3 $out/pytest_layouts.txt
(for rep in 1 2 3; do
  for lib in build/libntt_prev.so ""; do
    echo "rep $rep ${lib:-this build}: $(NTT_LIB=$lib timeout 300 python3 tools/pipeline_bench.py --logn 14 --batch 4096 | cut -c1-100)"
  done
  echo "rep $rep this build batch-major: $(timeout 300 python3 tools/pipeline_bench.py --logn 14 --batch 4096 --batch-major | cut -c1-120)"
done) > $out/rns_2p14_limb_major_regression.txt 2>&1
cat $out/rns_2p14_limb_major_regression.txt
Q=0x7fffffffe0001
(for rep in 1 2 3; do for o in 1 2 4 8 16; do
  echo "rep $rep config-2 shape (65536 x 2^12) oversub $o: $(timeout 200 python3 tools/sweep.py --qs 0x3ffffffffc001 --bytes 2147483648 --steps 20 --logn 12 --ops fwd inv --oversub $o | tail -n +2 | awk '{printf "%s:%s ", $4, $8}')"
done; done) > $out/oversub_config2_shape.txt 2>&1
cat $out/oversub_config2_shape.txt
timeout 2400 python3 -m pytest tests -m gpu -x -q > $out/pytest_gpu.txt 2>&1; tail -5 $out/pytest_gpu.txt
