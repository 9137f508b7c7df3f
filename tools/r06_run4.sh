#!/bin/bash
# fourth GPU call of round 6: config 5's bytes by item type, memset repro on torch's runtime, inverse one-pass prefetch position, full GPU suite
out=gpurun_out/r06; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 1500 bash tools/config5_bytes_by_item.sh $out/config5_items > $out/config5_bytes_by_item.txt 2>&1; cat $out/config5_bytes_by_item.txt
rm -rf $out/config5_items/*/*/*/*agent_info.csv $out/config5_items/*/*/*/*kernel_trace.csv 2>/dev/null
TL=$(python3 -c 'import os,torch;print(os.path.join(os.path.dirname(torch.__file__),"lib"))' 2>/dev/null)
(echo "# the same binary with the HIP runtime the torch wheel bundles preloaded (LD_PRELOAD=$TL/libamdhip64.so): the failing library test of round 5 ran on THAT runtime (torch imported first)"
 LD_PRELOAD=$TL/libamdhip64.so timeout 300 build/memset_graph_repro 12) > $out/memset_graph_repro_torch_runtime.txt 2>&1
head -3 $out/memset_graph_repro_torch_runtime.txt; grep -c "replays wrong" $out/memset_graph_repro_torch_runtime.txt; grep "replays wrong" $out/memset_graph_repro_torch_runtime.txt | grep -v " 0 of 12 replays" | head -20
Q=0x80000001c0001
(echo "# inverse one-pass 2^15: where the next polynomial's first half is requested (pair index of the final stage; 16 = behind it: the tree), alternating"
 for rep in 1 2; do
  echo "== behind the stage (tree)"; timeout 300 python3 tools/sweep.py --logn 15 --ops inv --qs $Q --bytes 16e9 | tail -1
  for at in 8 0; do echo "== in front of pair $at"; NTT_LIB=build/libntt_invpf$at.so timeout 300 python3 tools/sweep.py --logn 15 --ops inv --qs $Q --bytes 16e9 | tail -1; done
 done) > $out/onepass_inverse_prefetch_position.txt 2>&1
cat $out/onepass_inverse_prefetch_position.txt
timeout 1500 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -8
