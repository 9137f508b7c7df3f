#!/bin/bash
# tools/run_gpu_round.sh OUTDIR : GPU test suite + bench lines + small-batch RNS rows
out=$1; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 3000 python3 -m pytest tests -m gpu -x -q > $out/pytest_gpu.txt 2>&1
tail -5 $out/pytest_gpu.txt
(for lg in 14 16; do for limbs in 4 16; do for b in 1 2 8 64; do
  for loop in 1 0; do NTT_RNS_LOOP=$loop timeout 120 python3 tools/pipeline_bench.py --logn $lg --limbs $limbs --batch $b --steps 10; done
done; done; done) > $out/pipeline_rns_small_batch.txt 2>&1
cat $out/pipeline_rns_small_batch.txt
bash tools/run_bench_all.sh $out/bench
