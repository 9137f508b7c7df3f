#!/bin/bash
# first GPU call of round 6: memset-node reproducer, single-pass 2^15 skeleton, folded 8-shard rehearsal, bench tests
out=gpurun_out/r06; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 300 build/memset_graph_repro 12 > $out/memset_graph_repro.txt 2>&1; echo "memset repro rc $?"
tail -3 $out/memset_graph_repro.txt
timeout 300 build/skel15 16 16 > $out/skel15.txt 2>&1; echo "skel15 rc $?"
cat $out/skel15.txt
timeout 1500 bash tools/folded_8_shards.sh $out/folded_8_shards.txt
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -k "bench" 2>&1 | tail -5
